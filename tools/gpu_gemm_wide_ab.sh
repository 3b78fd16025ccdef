#!/bin/bash
# the wide GEMM form against the 128x128 kernel: equality test, then the headline step with both
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "wide_gemm or gemm_forms or gemm_modes or block_encoder or basis_encoder" 2>&1 | tail -4
for w in 0 1 0 1; do
  RGCN_GEMM_WIDE=$w timeout 300 python bench.py --steps 40 --warmup 8 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > gpurun_out/wide_$w.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("bench_details.json"))
print("RGCN_GEMM_WIDE=$w: %.4f ms/step  sum-exclusive %.4f ms" % (d["ms_per_step"], d["step_roofline"]["sum_exclusive_kernel_ms"]))
for k in d["kernels"]:
    if k["kernel"].startswith("gemm"): print("      %-18s x%.0f %7.1f us (pipelined %7.1f) frac %.3f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["frac"]))
PY
done
