#!/bin/bash
# stream schedules of the encoder step (form 3): side streams on / off, backward schedules
TAG=${1:-sched}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 50 --warmup 10 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile > gpurun_out/sched_${TAG}_$name.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("gpurun_out/sched_${TAG}_$name.json").read().strip().splitlines()[-1])
print("%-28s %.4f ms/step  %.2f M edges/s" % ("$name", d["ms_per_step"], d["value"] / 1e6))
PY
}
for rep in 1 2; do
run default RGCN_NOP=1
run streams0 RGCN_STREAMS=0
run bwd_sched0 RGCN_BWD_SCHED=0
run bwd_sched1 RGCN_BWD_SCHED=1
run fuse0 RGCN_FUSE=0
run fuse0_streams0 RGCN_FUSE=0 RGCN_STREAMS=0
done 2>&1 | tee gpurun_out/sched_$TAG.txt
