#!/bin/bash
# round 5: relation-ordered rows + coefficient reuse in k_block_rows -- prep / equality / parity subset, then the block workloads
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capture.py -m gpu -q --timeout 1200 -p no:cacheprovider -x \
  -k "graph_prep or two_kernel_form or full_graph_scale or block_encoder or golden or bias_gradient or deterministic or prefetched or sharding or sharded or training_graph_elementwise_parity or wn18_training or fb15k_training or block_edge or capture" > gpurun_out/pytest_r5_rows.log 2>&1
echo "pytest exit $?"; tail -n 8 gpurun_out/pytest_r5_rows.log
for rep in 1 2; do
for w in fb237_block fb237_block_fullgraph fb237_block_traingraph fb15k_block wn18_block; do
  timeout 600 python bench.py --workload $w --no-extra-workloads --steps 20 --warmup 5 --cpu-steps 0 --no-live-traffic --no-fp32-reference 2>/dev/null | tail -1 > gpurun_out/bench_rows_$w.json
  python - <<PY
import json
o = json.loads(open("gpurun_out/bench_rows_$w.json").read())
d = json.load(open(o["details"]))
ks = {k["kernel"]: k["avg_us"] for k in d["kernels"]}
print("%-24s %.4f ms/step   rows_fwd %.1f rows_bwd %.1f dw_msgs %.1f prep_sort %.1f" % (o["config"]["workload"], o["ms_per_step"], ks.get("block_rows_fwd", 0), ks.get("block_rows_bwd", 0), ks.get("block_dw_msgs", 0), ks.get("prep_sort", 0)))
PY
done; done
