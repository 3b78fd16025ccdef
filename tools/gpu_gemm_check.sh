#!/bin/bash
# GEMM-focused GPU visit: the dense-contraction tests first (fast, fail early), a timing table of the self-loop
# shapes, then the whole suite and the default bench.
TAG=${1:-gemm}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm" -p no:cacheprovider > gpurun_out/pytest_gemm_$TAG.log 2>&1
echo "gemm tests exit $?"; tail -n 15 gpurun_out/pytest_gemm_$TAG.log
timeout 300 python tools/gemm_time.py > gpurun_out/gemm_time_$TAG.log 2>&1; echo "gemm_time exit $?"; cat gpurun_out/gemm_time_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q -n 1 --timeout 600 -p no:cacheprovider "$@" > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?"
tail -n 30 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $?"
python tools/bench_table.py gpurun_out/bench_$TAG.json
tail -n 5 gpurun_out/bench_$TAG.err
