#!/bin/bash
# throw-away builds with kLongRow = 32 (shipped) / 64 / 128: combine at minibatch and at full-graph scale
export TMPDIR=/tmp
for L in 32 64 128; do
  sed -i "s/^constexpr int kLongRow = [0-9]*;/constexpr int kLongRow = $L;/" relationprediction_amd/csrc/rgcn_internal.h
  python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error" | head -3
  for w in fb237_block fb237_block_traingraph; do
    python bench.py --workload $w --steps 20 --warmup 5 --no-extra-workloads --cpu-steps 0 --no-fp32-reference 2>/dev/null | tail -1 > gpurun_out/lr.json
    python - <<PY
import json
c=json.loads(open("gpurun_out/lr.json").read()); d=json.load(open(c["details"]))
ks={k["kernel"]:k for k in d["kernels"]}
print("kLongRow=$L %-24s %.4f ms/step  combine_fwd %.1f  combine_bwd %.1f"%("$w", c["ms_per_step"], ks["combine_fwd"]["avg_us"], ks["combine_bwd"]["avg_us"]))
PY
  done
done
sed -i "s/^constexpr int kLongRow = [0-9]*;/constexpr int kLongRow = 32;/" relationprediction_amd/csrc/rgcn_internal.h
