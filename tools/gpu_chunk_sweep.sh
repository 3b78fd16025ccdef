#!/bin/bash
# messages per relation chunk (= per message-kernel workgroup) against the step: RGCN_CHUNK sweep on the headline workload
export TMPDIR=/tmp
mkdir -p gpurun_out
for ch in 16 24 32 48 64 96; do
  RGCN_CHUNK=$ch timeout 200 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference > gpurun_out/chunk_$ch.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/chunk_$ch.json").read().strip().splitlines()[-1])
k = {x["kernel"]: x for x in d["kernels"]}
print("RGCN_CHUNK=%-3d step %.3f ms  msg_fwd %.1f (%.1f)  msg_bwd %.1f (%.1f)  dw_reduce %.1f" % (
    $ch, d["ms_per_step"], k["block_msg_fwd"]["avg_us"], k["block_msg_fwd"]["avg_us_in_pipeline"],
    k["block_msg_bwd"]["avg_us"], k["block_msg_bwd"]["avg_us_in_pipeline"], k["block_dw_reduce"]["avg_us"]))
PY
done
