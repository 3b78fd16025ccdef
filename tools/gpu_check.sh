#!/bin/bash
# One GPU-box visit: parity tests (survive a crashing test via an xdist worker), then a short bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" 
timeout 1500 python -m pytest tests -m gpu -q -n 1 --timeout 600 -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"
tail -n 80 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?"
tail -c 6000 gpurun_out/bench.log; tail -n 20 gpurun_out/bench.err
