#!/bin/bash
# One GPU-box visit for a subset of the GPU tests: tools/gpu_pytest.sh TAG <pytest args...>
TAG=${1:-sub}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest -m gpu -q --timeout 900 -p no:cacheprovider "$@" > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"
tail -n 60 gpurun_out/pytest_$TAG.log
