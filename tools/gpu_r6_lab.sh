#!/bin/bash
TAG=${1:-a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/gemm_w8_lab.py 2>&1 | tee gpurun_out/gemm_w8_lab_$TAG.txt
