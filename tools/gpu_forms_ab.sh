#!/bin/bash
# the block layer's forms 0 / 2 / 3 side by side on the headline workload and on the 272,115-edge training graph
TAG=${1:-forms}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for wl in fb237_block fb237_block_traingraph; do for f in 0 2 3; do
  RGCN_FUSE=$f timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > gpurun_out/forms_${TAG}_${wl}_$f.json 2> gpurun_out/forms_${TAG}_${wl}_$f.err
  python - <<PY
import json
d = json.load(open("bench_details.json"))
print("RGCN_FUSE=$f $wl: %.4f ms/step  %.2f M edges/s  sum-exclusive %.4f ms" % (d["ms_per_step"], d["value"] / 1e6, d["step_roofline"]["sum_exclusive_kernel_ms"]))
for k in d["kernels"]:
    if k["kernel"].startswith(("block", "combine")): print("      %-18s x%.0f %7.1f us (pipelined %7.1f)" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"]))
PY
done; done 2>&1 | tee gpurun_out/forms_$TAG.txt
