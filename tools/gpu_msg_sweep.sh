#!/bin/bash
# workgroup size of the block message kernels (RGCN_MSG_BLOCK: 128 -> 1 slot, 256 -> 2, 512 -> 5 slots of 100 lanes) x
# messages per chunk (RGCN_CHUNK), headline workload
export TMPDIR=/tmp
mkdir -p gpurun_out
for blk in 128 256 320 512; do for ch in 16 24 48; do
  RGCN_MSG_BLOCK=$blk RGCN_CHUNK=$ch timeout 200 python bench.py --steps 30 --warmup 8 --cpu-steps 0 --no-extra-workloads --no-fp32-reference > gpurun_out/msgsweep.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/msgsweep.json").read().strip().splitlines()[-1])
k = {x["kernel"]: x for x in d["kernels"]}
print("RGCN_MSG_BLOCK=%-3d RGCN_CHUNK=%-3d step %.3f ms  msg_fwd %.1f (%.1f)  msg_bwd %.1f (%.1f)  dw_reduce %.1f (%.1f)" % (
    $blk, $ch, d["ms_per_step"], k["block_msg_fwd"]["avg_us"], k["block_msg_fwd"]["avg_us_in_pipeline"],
    k["block_msg_bwd"]["avg_us"], k["block_msg_bwd"]["avg_us_in_pipeline"], k["block_dw_reduce"]["avg_us"], k["block_dw_reduce"]["avg_us_in_pipeline"]))
PY
done; done
