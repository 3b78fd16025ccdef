#!/bin/bash
# per-wavefront timeline of the decoder's line kernel: needs tools/experiments/_libs/librgcn_trace.so
# (python tools/build_variant_libs.py decoder.hip trace=-DDEC_TRACE with the instrumented source)
mkdir -p gpurun_out; export TMPDIR=/tmp
cp relationprediction_amd/lib/librgcn.so /tmp/librgcn_product.so
cp tools/experiments/_libs/librgcn_trace.so relationprediction_amd/lib/librgcn.so
rm -f /tmp/dec_trace.bin
RGCN_DEC_TRACE=/tmp/dec_trace.bin RGCN_DEC_SPLIT=${SPLIT:-0} timeout 300 python tools/train_step_trace.py 4 > /dev/null 2>&1
python tools/dec_trace.py /tmp/dec_trace.bin | tee gpurun_out/dec_trace.txt
cp /tmp/librgcn_product.so relationprediction_amd/lib/librgcn.so
