#!/bin/bash
# the headline step twice (product library): per-kernel exclusive durations and the fabric traffic measured in the run
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
timeout 600 python bench.py --no-extra-workloads --steps 50 --warmup 10 --cpu-steps 0 --no-fp32-reference > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
d = json.load(open(o["details"]))
print("rep $rep", o["config"]["workload"], o["ms_per_step"], "ms/step   " + "  ".join("%s %.1f us %s MB" % (k["kernel"], k["avg_us"], "%.0f" % (k["traffic"] / 1e6) if k.get("traffic") else "-") for k in d["kernels"][:8]))
print("   step pmc bytes %.0f MB" % (d["step_roofline"]["pmc_bytes_per_step"] / 1e6))
PY
done
