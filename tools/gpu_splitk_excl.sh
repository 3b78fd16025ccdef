#!/bin/bash
# split-K target of the dW_self GEMM: step time AND the GEMM's exclusive duration (bench.py's per-kernel pass)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2; do for t in ${TARGETS:-256 320 384 448 512}; do
  RGCN_SPLITK_TARGET=$t timeout 300 python bench.py --steps 50 --warmup 10 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {x['kernel']: x for x in json.load(open('bench_details.json'))['kernels']}
print('target $t  %.4f ms/step   gemm_self_dw %.1f us alone (%.1f in the pipeline), splitk_reduce %.1f, roofline kernel %s %.4f' % (d['ms_per_step'], k['gemm_self_dw']['avg_us'], k['gemm_self_dw']['avg_us_in_pipeline'], k['splitk_reduce']['avg_us'], d['roofline']['kernel'], d['roofline']['frac']))"
done; done | tee gpurun_out/splitk_excl.txt
