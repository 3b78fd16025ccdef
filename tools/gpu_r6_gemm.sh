#!/bin/bash
# round 6 GEMM visit: bitwise tests of the pre-split-weight kernels, interleaved timing
TAG=${1:-a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "presplit" -p no:cacheprovider > gpurun_out/pytest_r6_gemm_$TAG.log 2>&1
echo "pytest exit $?"; tail -n 15 gpurun_out/pytest_r6_gemm_$TAG.log
timeout 300 python tools/gemm_w8_time.py 2>&1 | tee gpurun_out/gemm_w8_time_$TAG.txt
