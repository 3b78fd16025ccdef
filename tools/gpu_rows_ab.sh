#!/bin/bash
# A/B of the block layer's forms on one box: the equality tests, then the headline workload (and the 272,115-edge
# training graph) with RGCN_FUSE=0 (message kernel + k_combine) and RGCN_FUSE=3 (destination-major banded single pass,
# block_rows.hip), each with its per-kernel table.
#   tools/gpu_rows_ab.sh TAG [notest] [extra "ENV=VAL ENV2=VAL" variants of fuse 3 ...]
TAG=${1:-rows}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
if [ "$1" != "notest" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "fused or single_pass or training_graph_elementwise or determinism or block_encoder or golden" > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"; tail -n 15 gpurun_out/pytest_$TAG.log
else shift; fi
run() {   # name, workload, env...
  local name=$1; shift
  local wl=$1; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  echo "== $name ($wl) rc=$?"
  cp bench_details.json gpurun_out/bench_${TAG}_${name}_details.json
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}_${name}_details.json"))
print("   %.4f ms/step  %.2f M edges/s   sum exclusive %.4f ms" % (d["ms_per_step"], d["value"] / 1e6, d["step_roofline"]["sum_exclusive_kernel_ms"]))
for k in d["kernels"]:
    print("      %-18s x%.0f %7.1f us (pipelined %7.1f) frac %.3f design %.0f GB/s" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["frac"], k["design_gbs"]))
PY
}
run fuse0 fb237_block RGCN_FUSE=0
run fuse3 fb237_block RGCN_FUSE=3
i=0
for v in "$@"; do i=$((i+1)); run fuse3_v$i fb237_block RGCN_FUSE=3 $v; echo "   ($v)"; done
run tg_fuse0 fb237_block_traingraph RGCN_FUSE=0
run tg_fuse3 fb237_block_traingraph RGCN_FUSE=3
