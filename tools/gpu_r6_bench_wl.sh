#!/bin/bash
# workloads x RGCN_GEMM_W8 codes (devtools library), interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for wl in ${WLS:-fb237_basis_b2 fb237_basis_b5 wn18_block fb15k_block}; do
for code in ${CODES:-0 1}; do
RGCN_LIBRARY=devtools RGCN_GEMM_W8=$code timeout 600 python bench.py --workload $wl --no-extra-workloads --steps 30 --warmup 8 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_wl.json 2> gpurun_out/bench_wl.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_wl.json").read().strip().splitlines()[-1])
d = json.load(open(o["details"]))
ks = sorted(d["kernels"], key=lambda k: -k["avg_us"] * k["launches_per_step"])
print("rep $rep code $code %-16s %.4f ms/step   " % (o["config"]["workload"], o["ms_per_step"]) + "  ".join("%s %.1f" % (k["kernel"], k["avg_us"]) for k in ks if "gemm" in k["kernel"]))
PY
done
done
done
