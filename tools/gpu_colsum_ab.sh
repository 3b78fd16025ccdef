#!/bin/bash
# throw-away builds: rows per workgroup of the column-sum pass (db_emb = column sums of dW_emb [V,d])
export TMPDIR=/tmp
for L in 32 64 128; do
  sed -i "s/^constexpr int kColRowsPerBlock = [0-9]*;/constexpr int kColRowsPerBlock = $L;/" relationprediction_amd/csrc/elementwise.hip
  python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error" | head -3
  for i in 1 2; do
    python bench.py --steps 30 --warmup 5 --no-extra-workloads --cpu-steps 0 --no-fp32-reference --no-live-traffic 2>/dev/null | tail -1 > gpurun_out/cs.json
    python - <<PY
import json
c=json.loads(open("gpurun_out/cs.json").read()); d=json.load(open(c["details"]))
ks={k["kernel"]:k for k in d["kernels"]}
print("rows/block=$L  %.4f ms/step  bias_grad_colsum %.1f us (pipelined %.1f)"%(c["ms_per_step"], ks["bias_grad_colsum"]["avg_us"], ks["bias_grad_colsum"]["avg_us_in_pipeline"]))
PY
  done
done
sed -i "s/^constexpr int kColRowsPerBlock = [0-9]*;/constexpr int kColRowsPerBlock = 32;/" relationprediction_amd/csrc/elementwise.hip
