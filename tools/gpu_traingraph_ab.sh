#!/bin/bash
# the backward-layer schedule knobs of form 3 (RGCN_ROWS_SERIAL x RGCN_GEMM_CORUN) on one workload:
# WL=fb237_block_traingraph (default; 272,115 edges) or WL=fb237_block (the headline minibatch)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload ${WL:-fb237_block_traingraph} --steps ${STEPS:-20} --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4f ms/step' % ('$name', d['ms_per_step']))"; }
for rep in 1 2 3; do
  run serial1_corun1 RGCN_ROWS_SERIAL=1 RGCN_GEMM_CORUN=1; run serial1_corun0 RGCN_ROWS_SERIAL=1 RGCN_GEMM_CORUN=0
  run serial0_corun1 RGCN_ROWS_SERIAL=0 RGCN_GEMM_CORUN=1; run serial0_corun0 RGCN_ROWS_SERIAL=0 RGCN_GEMM_CORUN=0
done 2>&1 | tee gpurun_out/sched_${WL:-fb237_block_traingraph}.txt
