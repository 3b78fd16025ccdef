#!/bin/bash
# the device train step (prep + encoder + DistMult + clip + Adam): minibatch step, hipGraph replay, stream launches,
# under a few knobs.  tools/gpu_trainstep_ab.sh TAG ["ENV=VAL ..." ...]
TAG=${1:-ts}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for v in "RGCN_NOP=1" "$@"; do
  echo "== $v"
  env $v timeout 300 python tools/train_step_probe.py fb237_block_train_step 40 2>/dev/null | head -${LINES_SHOWN:-1}
done 2>&1 | tee gpurun_out/trainstep_$TAG.txt
