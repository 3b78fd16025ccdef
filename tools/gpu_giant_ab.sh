#!/bin/bash
# A/B: hub rows cut into pieces at minibatch scale (RGCN_GIANT_LEN): k_combine is bound by its longest row
TAG=${1:-giant}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  python - <<PY
import json
d = json.load(open("bench_details.json"))
k = {x["kernel"]: x for x in d["kernels"]}
print("%-12s step %.4f ms  sum %.4f | combine_fwd %5.1f (%5.1f)  combine_bwd %5.1f (%5.1f)  msg_bwd %5.1f  msg_fwd %5.1f" % ("$name", d["ms_per_step"], d["step_roofline"]["sum_exclusive_kernel_ms"], k["combine_fwd"]["avg_us"], k["combine_fwd"]["avg_us_in_pipeline"], k["combine_bwd"]["avg_us"], k["combine_bwd"]["avg_us_in_pipeline"], k["block_msg_bwd"]["avg_us"], k["block_msg_fwd"]["avg_us"]))
PY
}
run base
for L in 64 128 256; do run len$L RGCN_GIANT_LEN=$L; done
run base2
for L in 64 256; do
RGCN_GIANT_LEN=$L timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fused or determinism or block_encoder or golden" > gpurun_out/pytest_${TAG}_$L.log 2>&1; tail -2 gpurun_out/pytest_${TAG}_$L.log
done
