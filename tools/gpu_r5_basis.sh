#!/bin/bash
# Round 5: the row-compacted basis contraction -- parity subset, then the two basis workloads with kernel tables.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiprocess.py tests/test_gpu_plugin.py -m gpu -q --timeout 1200 -p no:cacheprovider -x \
  -k "basis or gemm_forms or float64 or golden or sharding or plugin" > gpurun_out/pytest_r5_basis.log 2>&1
echo "pytest exit $?"
tail -n 25 gpurun_out/pytest_r5_basis.log
for w in fb237_basis_b2 fb237_basis_b5; do
  timeout 600 python bench.py --workload $w --no-extra-workloads --steps 20 --warmup 5 --cpu-steps 0 --no-live-traffic --no-fp32-reference > gpurun_out/bench_r5_$w.json 2> gpurun_out/bench_r5_$w.err
  echo "bench $w exit $?"
  python - <<PY
import json
o = json.loads(open("gpurun_out/bench_r5_$w.json").read().strip().splitlines()[-1])
print(o["config"]["workload"], o["ms_per_step"], "ms/step", o["value"] / 1e6, "M edges/s")
d = json.load(open(o["details"]))
for k in d["kernels"]:
    print("   %-22s x%.0f %7.1f us (pipelined %7.1f) %s frac %.3f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["bound"], k["frac"]))
PY
done
