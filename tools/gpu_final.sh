#!/bin/bash
# what the driver does at round end, in one box visit: the GPU suite, smoke(), the default bench (cold)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "bench exit $?"; cp bench_details.json gpurun_out/bench_final_details.json
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1200 -p no:cacheprovider > gpurun_out/pytest_final.log 2>&1
echo "pytest exit $?"; tail -n 4 gpurun_out/pytest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(len(json.dumps(o)), "bytes;", o["config"]["workload"], o["ms_per_step"], o["value"], o["roofline"]["kernel"], o["roofline"]["frac"], o["roofline"]["avg_us"], o["roofline"]["traffic"])
print([(w["workload"], w["ms_per_step"]) for w in o["workloads"]])
print(o["train_steps"]); print(o["evaluation"]); print(o["evaluation_encodes"]); print(o["step_roofline"]); print(o["cpu_baseline"])
PY
