#!/bin/bash
# the default bench twice (in-process workload order effects), compact tables
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_bo$rep.json 2> gpurun_out/bench_bo$rep.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_bo$rep.json").read().strip().splitlines()[-1])
print("rep $rep", o["config"]["workload"], o["ms_per_step"], [(w["workload"], w["ms_per_step"]) for w in o["workloads"]])
print("   train", [(t["workload"], t["ms_per_step"], t["hipgraph_ms"], t["stream_ms"]) for t in o["train_steps"]], "eval", o["evaluation"]["encode_full_graph_ms"], o.get("evaluation_encodes"))
PY
done
