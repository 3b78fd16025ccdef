#!/bin/bash
# hipGraph replay of the train step: the chain rgcn_capture_begin records against the stream path's whole DAG, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for cs in 0 1; do
RGCN_LIBRARY=devtools RGCN_CAPTURE_STREAMS=$cs timeout 600 python tools/capture_ab.py 2>&1 | grep -v "^$" | tail -3
done
done | tee gpurun_out/capture_ab.txt
