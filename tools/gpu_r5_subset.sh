#!/bin/bash
# usage: tools/gpu_r5_subset.sh TAG "<pytest -k expression>" [pytest paths...]
TAG=$1; K=$2; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1800 python -m pytest ${@:-tests} -m gpu -q --timeout 1200 -p no:cacheprovider -k "$K" --durations=5 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"
tail -n 30 gpurun_out/pytest_$TAG.log
