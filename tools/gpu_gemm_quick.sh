#!/bin/bash
# quick GEMM visit: correctness of the dense-contraction tests, timing table, PMC passes
TAG=${1:-q}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm" -p no:cacheprovider 2>&1 | tail -n 12
timeout 300 python tools/gemm_time.py 2>&1 | tee gpurun_out/gemm_time_$TAG.log
bash tools/gpu_gemm_pmc.sh > gpurun_out/gemm_pmc_$TAG.log 2>&1
grep -A9 "k_gemm_planes" gpurun_out/gemm_pmc_$TAG.log | head -120
