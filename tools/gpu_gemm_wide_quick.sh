#!/bin/bash
# wide GEMM quick timing: headline step, per-kernel GEMM rows only
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for w in 1 0; do
  env RGCN_GEMM_WIDE=$w "$@" timeout 300 python bench.py --steps 40 --warmup 8 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > gpurun_out/wide_$w.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("bench_details.json"))
print("RGCN_GEMM_WIDE=$w: %.4f ms/step  sum-exclusive %.4f ms  " % (d["ms_per_step"], d["step_roofline"]["sum_exclusive_kernel_ms"]) + "  ".join("%s %.1f" % (k["kernel"], k["avg_us"]) for k in d["kernels"] if k["kernel"].startswith("gemm")))
PY
done
