#!/bin/bash
# throw-away builds of block_rows.hip with experiment knobs, each timed on the headline workload:
#   tools/gpu_rows_variants.sh TAG "<-D flags of variant 1>" "<-D flags of variant 2>" ...
TAG=${1:-v}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for v in "" "$@"; do
  RGCN_EXTRA_HIPCC_FLAGS="$v $BASEFLAGS" python -c "from relationprediction_amd import build; build.build()" 2>&1 | grep -v "not a recognized" | tail -2
  for wl in ${WLS:-fb237_block fb237_block_traingraph}; do
  env RGCN_FUSE=3 $RUNENV timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > gpurun_out/var_${TAG}_$i.json 2> gpurun_out/var_${TAG}_$i.err
  python - <<PY
import json
d = json.load(open("bench_details.json"))
ks = {k["kernel"]: k for k in d["kernels"]}
print("variant $i [$v] $wl: %.4f ms/step sum-excl %.4f  rows_fwd %.1f (pipe %.1f) rows_bwd %.1f (pipe %.1f)" % (d["ms_per_step"], d["step_roofline"]["sum_exclusive_kernel_ms"], ks["block_rows_fwd"]["avg_us"], ks["block_rows_fwd"]["avg_us_in_pipeline"], ks["block_rows_bwd"]["avg_us"], ks["block_rows_bwd"]["avg_us_in_pipeline"]))
PY
  done
  i=$((i+1))
done
RGCN_EXTRA_HIPCC_FLAGS="" python -c "from relationprediction_amd import build; build.build()" 2>&1 | tail -1
