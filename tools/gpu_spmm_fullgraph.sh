#!/bin/bash
# the single-pass block SpMM against the two-kernel form at full-graph scale (38 k and 272 k edges) and in the
# evaluation encode
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
for w in fb237_block_fullgraph fb237_block_traingraph; do
for f in 0 2; do
  RGCN_FUSE=$f timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --cpu-steps 0 --no-extra-workloads --no-fp32-reference > gpurun_out/fg_$w_$f.json 2> gpurun_out/fg_$w_$f.err
  python - <<PY
import json
d = json.load(open("bench_details.json"))
print("$w fuse=$f  %.4f ms/step  %.1f M edges/s" % (d["ms_per_step"], d["value"] / 1e6))
for k in d["kernels"][:8]:
    print("      %-18s x%.0f %8.1f us" % (k["kernel"], k["launches_per_step"], k["avg_us"]))
PY
done; done
