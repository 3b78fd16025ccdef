#!/bin/bash
# split-K of the dW_self GEMM (RGCN_SPLITK_TARGET workgroups) now that it runs beside the dH GEMM
#   WL=<workload> TARGETS="256 512" tools/gpu_splitk_ab.sh
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload ${WL:-fb237_block} --steps 50 --warmup 10 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4f ms/step' % ('$name', d['ms_per_step']))"; }
for rep in 1 2 3; do for t in ${TARGETS:-256 384 512 768}; do run ${WL:-fb237_block}_target_$t RGCN_SPLITK_TARGET=$t; done; done | tee gpurun_out/splitk_ab_${WL:-fb237_block}.txt
