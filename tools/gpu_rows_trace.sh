#!/bin/bash
# per-wavefront timeline of k_block_rows (diagnostic): tools/gpu_rows_trace.sh TAG [ENV=VAL ...]
TAG=${1:-t}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
rm -f /tmp/rows_trace.bin
env RGCN_FUSE=3 RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0 RGCN_ROWS_TRACE=/tmp/rows_trace.bin "$@" timeout 300 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile > gpurun_out/trace_$TAG.json 2> gpurun_out/trace_$TAG.err
for k in -4 -3 -2 -1; do python tools/rows_trace.py /tmp/rows_trace.bin $k; done | tee gpurun_out/trace_$TAG.txt
