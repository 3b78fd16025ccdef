#!/bin/bash
# A/B of the backward schedule of the single-pass block layer (form 3): RGCN_GEMM_CORUN=1 (default: the slab reduce of
# the relation-weight gradients and dW_self = H^T.dS on side stream 1 beside dH = dS.W^T) against 0 (one behind the
# other).  Run on the GPU box:  gpurun -- 'bash tools/gpu_corun_ab.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 50 --warmup 10 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4f ms/step' % ('$name', d['ms_per_step']))"; }
for rep in 1 2 3; do run corun_on RGCN_GEMM_CORUN=1; run corun_off RGCN_GEMM_CORUN=0; done
