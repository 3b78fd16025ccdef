#!/bin/bash
# A/B of the block layer's forms on one box: the equality tests, then the headline workload with RGCN_FUSE=0 (message
# kernel + k_combine) and RGCN_FUSE=2 (single-pass block SpMM), each with its per-kernel table; optional chunk sweep.
#   tools/gpu_spmm_ab.sh TAG [chunks...]
TAG=${1:-spmm}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fused or single_pass or block_encoder or determinism or golden" > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"; tail -n 15 gpurun_out/pytest_$TAG.log
run() {   # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  echo "== $name rc=$?"
  cp bench_details.json gpurun_out/bench_${TAG}_${name}_details.json
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}_${name}_details.json"))
print("   %.4f ms/step  %.2f M edges/s   sum exclusive %.4f ms" % (d["ms_per_step"], d["value"] / 1e6, d["step_roofline"]["sum_exclusive_kernel_ms"]))
for k in d["kernels"]:
    print("      %-18s x%.0f %7.1f us (pipelined %7.1f) frac %.3f design %.0f GB/s" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["frac"], k["design_gbs"]))
PY
}
run fuse0 RGCN_FUSE=0
run fuse2 RGCN_FUSE=2
for ch in "$@"; do run fuse2_ch$ch RGCN_FUSE=2 RGCN_SPMM_CHUNKS=$ch; done
