#!/bin/bash
mkdir -p gpurun_out
for w in "$@"; do
  echo "== $w"
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-steps 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || tail -5 gpurun_out/bench_$w.err
  python - <<PY
import json
o = json.loads(open("gpurun_out/bench_$w.json").read().strip().splitlines()[-1])
print("ms/step %.4f  edges/s %.0f  cpu %s" % (o["ms_per_step"], o["value"], o["cpu_baseline"] and o["cpu_baseline"]["value"]))
for k in o["kernels"][:16]:
    print("   %-22s %5.1f/step  avg %8.2f us  %8.4f ms/step  %s %.3f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k["bound"], k["frac"]))
PY
done
