#!/bin/bash
TAG=${1:-ser}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 50 --warmup 10 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic --no-kernel-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4f ms/step' % ('$name', d['ms_per_step']))"; }
for rep in 1 2 3; do
run rows_serial1 RGCN_ROWS_SERIAL=1
run rows_serial0 RGCN_ROWS_SERIAL=0
run streams0 RGCN_STREAMS=0
done 2>&1 | tee gpurun_out/serial_$TAG.txt
for v in "RGCN_ROWS_SERIAL=1" "RGCN_ROWS_SERIAL=0"; do echo "== $v"; env $v python tools/train_step_probe.py fb237_block_train_step 40 2>/dev/null | head -1 | cut -c1-330; done
