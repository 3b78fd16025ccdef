#!/bin/bash
# the device-side collective stand-in (tests/collective_double/ipc_collective.hip): first under the ordinary sharded
# worker (steps issued directly), then the captured sharded train step
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
LIB=$PWD/tests/collective_double/_build/libipccollective.so
for w in 2 3; do
  RGCN_LIBRARY=devtools RGCN_RCCL_LIBRARY=$LIB timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 \
    --master-port $((29800 + w)) tests/collective_double/sharded_worker.py block 8 2>&1 | grep -v "elastic\|torch/distributed\|^\*\*\*\|OMP_NUM" | tail -8
done
RGCN_CAPTURE_SHARDED=1 RGCN_LIBRARY=devtools RGCN_RCCL_LIBRARY=$LIB timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29810 tests/collective_double/captured_worker.py block 8 2>&1 | grep -v "elastic\|torch/distributed\|^\*\*\*\|OMP_NUM" | tail -12
