#!/bin/bash
# headline + the 272,115-edge step, three repetitions each (box-internal spread)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2 3; do
for w in fb237_block fb237_block_traingraph; do
  timeout 600 python bench.py --workload $w --no-extra-workloads --steps 20 --warmup 5 --cpu-steps 0 --no-live-traffic --no-fp32-reference --no-kernel-profile 2>/dev/null | tail -1 | python -c "import json,sys; o=json.loads(sys.stdin.read()); print(o['config']['workload'], o['ms_per_step'])"
done; done
