#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_step.py tests/test_gpu_capture.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "basis" 2>&1 | tail -2
run() { name=$1; wl=$2; shift; shift; env "$@" timeout 300 python bench.py --workload $wl --steps 30 --warmup 6 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %.4f ms/step' % ('$name $wl', d['ms_per_step']))"; }
for rep in 1 2; do for wl in fb237_basis_b2 fb237_basis_b5; do run corun0 $wl RGCN_GEMM_CORUN=0; run corun1 $wl RGCN_GEMM_CORUN=1; done; done
