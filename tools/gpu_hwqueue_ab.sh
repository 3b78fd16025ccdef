#!/bin/bash
# Is the intermittent 2x slowdown of a LATER engine's stream-launched train step (same kernels, GPU idle half the time)
# a stream -> hardware-queue mapping effect?  Full default bench three times as is, three times with 8 hardware queues.
export TMPDIR=/tmp
mkdir -p gpurun_out
for q in "" 8; do
  for i in 1 2 3; do
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-fp32-reference > gpurun_out/hwq_${q:-default}_$i.json 2> gpurun_out/hwq_${q:-default}_$i.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/hwq_${q:-default}_$i.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES=${q:-default} run $i: headline %.3f ms" % d["ms_per_step"],
      " ".join("%s %.3f" % (w["config"]["workload"][6:], w["ms_per_step"]) for w in d["workloads"]),
      " | train steps (minibatch / graph / streams):",
      " ".join("%.3f/%.3f/%.3f" % (t["minibatch_step"]["ms_per_step"], t["captured_step"]["ms_per_step_hipgraph_replay"],
                                   t["captured_step"]["ms_per_step_stream_launched"]) for t in d["train_steps"]))
PY
  done
done
