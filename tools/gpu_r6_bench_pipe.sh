#!/bin/bash
# exclusive and pipelined kernel durations of a workload under two RGCN_GEMM_W8 codes
export TMPDIR=/tmp
mkdir -p gpurun_out
WL=${WL:-fb237_block}
for code in ${CODES:-0 1}; do
RGCN_LIBRARY=devtools RGCN_GEMM_W8=$code timeout 600 python bench.py --workload $WL --no-extra-workloads --steps 50 --warmup 10 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_pipe_$code.json 2> gpurun_out/bench_pipe_$code.err
cp bench_details.json gpurun_out/bench_pipe_details_$code.json
done
python - <<PY
import json
codes = "${CODES:-0 1}".split()
ds = {c: json.load(open("gpurun_out/bench_pipe_details_%s.json" % c)) for c in codes}
names = [k["kernel"] for k in ds[codes[0]]["kernels"]]
print("%-22s" % "$WL" + "".join("   code %s: ms/step %.4f        " % (c, ds[c]["ms_per_step"]) for c in codes))
for n in names:
    row = "%-22s" % n
    for c in codes:
        k = [k for k in ds[c]["kernels"] if k["kernel"] == n]
        row += "   x%.0f excl %6.1f  piped %6.1f   " % (k[0]["launches_per_step"], k[0]["avg_us"], k[0]["avg_us_in_pipeline"]) if k else "   -"
    print(row)
PY
