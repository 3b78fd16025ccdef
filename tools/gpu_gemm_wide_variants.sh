#!/bin/bash
# throw-away builds of the wide GEMM with ablation knobs (GW_ABL bits: 1 no fragment reads, 2 no A loads, 4 no B loads,
# 8 no A split/store, 16 no B store, 32 no barrier): WRONG results, timing only
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
  RGCN_EXTRA_HIPCC_FLAGS="$v" python -c "from relationprediction_amd import build; build.build()" 2>&1 | grep -v "not a recognized" | tail -1
  RGCN_GEMM_WIDE=1 timeout 300 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > /dev/null 2>&1
  python - <<PY
import json
d = json.load(open("bench_details.json"))
print("[$v] " + "  ".join("%s %.1f" % (k["kernel"], k["avg_us"]) for k in d["kernels"] if k["kernel"].startswith("gemm")))
PY
done
RGCN_EXTRA_HIPCC_FLAGS="" python -c "from relationprediction_amd import build; build.build()" 2>&1 | tail -1
