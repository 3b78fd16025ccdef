#!/bin/bash
# One GPU-box visit: parity tests (survive a crashing test via an xdist worker), then the default bench
# (headline workload + the "workloads" array).  Usage: tools/gpu_check.sh [tag] [extra pytest args]
TAG=${1:-check}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing"
timeout 1500 python -m pytest tests -m gpu -q -n 1 --timeout 1200 -p no:cacheprovider --durations=8 "$@" > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?"
tail -n 40 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $?"
python - <<PY
import json
last = open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1]
print("result line: %d bytes" % len(last))
print(last)
c = json.loads(last)
d = json.load(open(c["details"]))
def line(o):
    r = o.get("roofline") or {}
    return "%-24s %8.3f ms/step %10.2f M edges/s   top kernel %-16s %6.1f us frac %.3f" % (
        o["config"]["workload"], o["ms_per_step"], o["value"] / 1e6, r.get("kernel"), r.get("avg_us", 0), r.get("frac", 0))
print(line(d))
for o in d.get("workloads", []):
    print(line(o))
for t in d.get("train_steps", []):
    print("  ", t["workload"], {k: v for k, v in t.get("minibatch_step", {}).items() if k != "kernels"}, t.get("captured_step"))
    for k in t.get("minibatch_step", {}).get("kernels", [])[:8]:
        print("      %-22s x%.0f %7.1f us %s frac %.3f (design %.0f GB/s)" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["bound"], k["frac"], k["design_gbs"]))
print("cpu", d.get("cpu_baseline"))
print("step_roofline", {k: v for k, v in (d.get("step_roofline") or {}).items() if k != "note"})
for k in d["kernels"]:
    print("   %-22s x%.0f %7.1f us (pipelined %7.1f) %s frac %.3f  comp %.1f MB design %.1f MB pmc %s" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["bound"], k["frac"], k["compulsory_bytes"] / 1e6, k["design_bytes"] / 1e6, k["traffic"]))
PY
tail -n 20 gpurun_out/bench_$TAG.err
