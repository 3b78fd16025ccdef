#!/bin/bash
# generic A/B of environment knobs on one bench workload: WL=<workload> tools/gpu_env_ab.sh "A=1" "B=2 C=3" ...
# (the unmodified default runs first in every repetition)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { name="$1"; env $1 timeout 300 python bench.py --workload ${WL:-fb237_block} --steps ${STEPS:-50} --warmup 10 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f ms/step' % ('$name', d['ms_per_step']))"; }
for rep in 1 2 3; do run "RGCN_NOP=1"; for v in "$@"; do run "$v"; done; done | tee gpurun_out/env_ab.txt
