#!/bin/bash
# round 5: the headline step three times per setting of RGCN_EVFLAG (read by the throw-away build of the fence-scope A/B;
# the shipped library ignores it: three repeats), then the block / capture / train-step tests
#
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2 3; do
  for f in 0 1 2; do
    RGCN_EVFLAG=$f timeout 300 python bench.py --steps 50 --warmup 10 --cpu-steps 0 --no-kernel-profile --no-extra-workloads --no-fp32-reference > gpurun_out/evflag_$f.json 2> gpurun_out/evflag_$f.err
    python - <<PY
import json
o = json.loads(open("gpurun_out/evflag_$f.json").read().strip().splitlines()[-1])
print("rep $rep flag $f", o["ms_per_step"])
PY
  done
done
RGCN_EVFLAG=${1:-2} timeout 1500 python -m pytest tests -m gpu -x -q -k "block or capture or train_step or minibatch or prefetch" 2>&1 | tail -5
