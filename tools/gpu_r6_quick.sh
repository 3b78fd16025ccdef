#!/bin/bash
# the headline step twice (product library), per-kernel exclusive durations
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
timeout 600 python bench.py --no-extra-workloads --steps 50 --warmup 10 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
d = json.load(open(o["details"]))
print("rep $rep", o["config"]["workload"], o["ms_per_step"], "ms/step   " + "  ".join("%s %.1f" % (k["kernel"], k["avg_us"]) for k in d["kernels"]))
PY
done
