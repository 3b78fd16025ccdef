#!/bin/bash
# hipGraph replay against stream launches of the same steps (BASELINE config 5 names a captured train step)
mkdir -p gpurun_out; : > gpurun_out/hipgraph_ab.log
run() {
  echo "== $*" | tee -a gpurun_out/hipgraph_ab.log
  env $1 python bench.py --steps 40 --warmup 6 --cpu-steps 0 --no-extra-workloads --no-kernel-profile $2 2>&1 | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ms/step %.4f (gpu events %.4f)  edges/s %.0f' % (o['ms_per_step'], o['gpu_event_ms_per_step'], o['value']))
" | tee -a gpurun_out/hipgraph_ab.log
}
run "RGCN_X=0" ""
run "RGCN_X=0" "--hipgraph"
run "RGCN_STREAMS=0" ""
run "RGCN_STREAMS=0" "--hipgraph"
run "RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0" ""
run "RGCN_BENCH_GRAPH_NOPF=1" "--hipgraph"
run "RGCN_STREAMS=0 RGCN_BENCH_GRAPH_NOPF=1" "--hipgraph"
