#!/bin/bash
# Round 5, first visit: the new parity tests (basis B = 5, 5,000-wide GEMMs, the full-graph encodes) on the tree as round 4
# left it, then the default bench for this round's reference numbers.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
nproc
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 1200 -p no:cacheprovider --durations=12 \
  -k "basis or gemm_forms or training_graph or float64 or wn18 or fb15k" > gpurun_out/pytest_r5_first.log 2>&1
echo "pytest exit $?"
tail -n 40 gpurun_out/pytest_r5_first.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5_first.json 2> gpurun_out/bench_r5_first.err
echo "bench exit $?"
tail -c 6000 gpurun_out/bench_r5_first.json
