#!/bin/bash
# the short-row prototype kernel (RGCN_ROWS2=1, devtools library): bitwise tests, then the headline step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
RGCN_LIBRARY=devtools RGCN_ROWS2=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "single_pass or block_encoder or fb237_minibatch_full or float64_tie or bitwise_determ" 2>&1 | tail -5
for rep in 1 2; do
for r2 in 0 1; do
RGCN_LIBRARY=devtools RGCN_ROWS2=$r2 timeout 600 python bench.py --no-extra-workloads --steps 50 --warmup 10 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_rows2_$r2.json 2> gpurun_out/bench_rows2_$r2.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_rows2_$r2.json").read().strip().splitlines()[-1])
d = json.load(open(o["details"]))
ks = {k["kernel"]: k for k in d["kernels"]}
print("rep $rep rows2=$r2", o["config"]["workload"], o["ms_per_step"], "ms/step   " + "  ".join("%s x%.0f %.1f (piped %.1f)" % (n, ks[n]["launches_per_step"], ks[n]["avg_us"], ks[n]["avg_us_in_pipeline"]) for n in ("block_rows_fwd", "block_rows_bwd", "gemm_self_fwd") if n in ks))
PY
done
done
