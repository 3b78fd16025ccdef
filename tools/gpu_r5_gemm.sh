#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "presplit or gemm_forms or basis_encoder or block_encoder or golden or float64" > gpurun_out/pytest_r5_gemm.log 2>&1
echo "pytest exit $?"; tail -n 12 gpurun_out/pytest_r5_gemm.log
timeout 300 python tools/gemm_presplit_time.py 2>&1 | tee gpurun_out/gemm_presplit_time.txt
timeout 600 python bench.py --no-extra-workloads --steps 20 --warmup 5 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_r5_gemm.json 2> gpurun_out/bench_r5_gemm.err
python - <<PY
import json
o = json.loads(open("gpurun_out/bench_r5_gemm.json").read().strip().splitlines()[-1])
print(o["config"]["workload"], o["ms_per_step"], "ms/step")
d = json.load(open(o["details"]))
for k in d["kernels"]:
    print("   %-22s x%.0f %7.1f us (pipelined %7.1f) %s frac %.3f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["bound"], k["frac"]))
PY
