#!/bin/bash
# the headline step with the devtools library under different RGCN_GEMM_W8 codes, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
CODES=${CODES:-"0 1 2"}
for rep in 1 2; do
for code in $CODES; do
RGCN_LIBRARY=devtools RGCN_GEMM_W8=$code timeout 600 python bench.py --no-extra-workloads --steps 50 --warmup 10 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_ab_$code.json 2> gpurun_out/bench_ab_$code.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_ab_$code.json").read().strip().splitlines()[-1])
d = json.load(open(o["details"]))
ks = {k["kernel"]: k for k in d["kernels"]}
print("rep $rep code $code", o["config"]["workload"], o["ms_per_step"], "ms/step   " + "  ".join("%s %.1f" % (n, ks[n]["avg_us"]) for n in ("gemm_self_fwd", "gemm_self_dh", "gemm_self_dw", "block_rows_fwd", "block_rows_bwd") if n in ks))
PY
done
done
