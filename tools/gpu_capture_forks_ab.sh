#!/bin/bash
# which forks pay inside a CAPTURED train step (RGCN_CAPTURE_FORKS: bit 1 = the backward layer's GEMM pairing, bit 2 =
# the decoder's relation-gradient reduce; the decoder's preparation is always forked): replayed against stream-launched
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2 3; do for v in ${FORKS:-0 1 2 3}; do
  RGCN_CAPTURE_FORKS=$v timeout 300 python tools/train_step_probe.py ${WL:-fb237_block_train_step} 40 2>/dev/null | head -1 | python -c "
import sys, re
l = sys.stdin.read(); m = re.search(r\"hipgraph_replay': ([0-9.]+), 'ms_per_step_stream_launched': ([0-9.]+)\", l)
print('forks=$v  replay %s  stream %s' % (m.group(1), m.group(2)))"
done; done | tee gpurun_out/capture_forks_${WL:-fb237_block_train_step}.txt
