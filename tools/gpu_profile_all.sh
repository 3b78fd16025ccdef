#!/bin/bash
# rocprofv3 summaries (kernel trace + FETCH_SIZE / WRITE_SIZE passes) of every bench workload and of the device train
# step, one box visit: gpurun_out/<TAG>_rocprof_*.md / *_traffic.json -- copy into profiles/.
TAG=${1:-r03}
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
bash tools/gpu_profile.sh $TAG fb237_block | tail -3
for w in fb237_block fb237_basis_b2 fb237_basis_b5 wn18_block fb15k_block fb237_block_fullgraph fb237_block_traingraph; do
  echo "== $w serial"; bash tools/gpu_profile.sh $TAG $w serial | grep -E "rc=|total GPU" ; done
bash tools/gpu_profile_train.sh $TAG | grep -E "rc=|total GPU"
