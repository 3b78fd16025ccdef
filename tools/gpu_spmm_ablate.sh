#!/bin/bash
# Where the single-pass block kernel's time goes: the headline workload with parts of the kernel switched off
# (RGCN_SPMM_ABLATE bits: 1 no row sums, 2 no gathers, 4 no weight-table reads, 8 no epilogue stores; results are then
# wrong, only the durations matter), the two workgroup orders and a few chunk counts.
TAG=${1:-ablate}; shift
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fused or single_pass" > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"; tail -n 3 gpurun_out/pytest_$TAG.log
run() {
  local name=$1; shift
  env RGCN_FUSE=2 "$@" timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  python - <<PY
import json
d = json.load(open("bench_details.json"))
k = {x["kernel"]: x for x in d["kernels"]}
print("%-28s step %.4f ms   spmm_fwd %6.1f us   spmm_bwd %6.1f us   dw_msgs %5.1f us" % ("$name", d["ms_per_step"], k["block_spmm_fwd"]["avg_us"], k["block_spmm_bwd"]["avg_us"], k.get("block_dw_msgs", {}).get("avg_us", 0)))
PY
}
run base
for ab in 1 2 4 8 3 7 15; do run ablate$ab RGCN_SPMM_ABLATE=$ab; done
run order0 RGCN_SPMM_ORDER=0
run order0_ch9 RGCN_SPMM_ORDER=0 RGCN_SPMM_CHUNKS=9
run order1_ch9 RGCN_SPMM_ORDER=1 RGCN_SPMM_CHUNKS=9
run order1_ch32 RGCN_SPMM_ORDER=1 RGCN_SPMM_CHUNKS=32
