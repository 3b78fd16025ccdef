#!/bin/bash
# Time a kernel of the device train step under the variant libraries of tools/build_variant_libs.py (built locally,
# shipped with the tree): tools/gpu_variant_libs.sh KERNEL_TAG [variant ...]   (default: every library found)
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-dec_entity_grad}; shift
LIBS=${@:-$(ls tools/experiments/_libs/ | sed -n 's/^librgcn_\(.*\)\.so$/\1/p')}
cp relationprediction_amd/lib/librgcn.so /tmp/librgcn_product.so
for v in $LIBS; do
  cp tools/experiments/_libs/librgcn_$v.so relationprediction_amd/lib/librgcn.so
  timeout 300 python tools/train_step_probe.py fb237_block_train_step 30 2>/dev/null | awk -v v=$v -v t=$TAG 'NR==1{split($0,a,"ms_per_step\x27: "); split(a[2],b,","); ms=b[1]} $1==t{printf "%-14s %s %s us   step %s ms\n", v, $1, $3, ms}'
done | tee gpurun_out/variant_libs_$TAG.txt
cp /tmp/librgcn_product.so relationprediction_amd/lib/librgcn.so
