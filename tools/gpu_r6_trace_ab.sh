#!/bin/bash
# kernel timeline of the headline step with the devtools library under two RGCN_GEMM_W8 codes
export TMPDIR=/tmp
ROOTDIR=$GRAFT_REPO_ROOT
for code in ${CODES:-0 1}; do
OUT=$ROOTDIR/gpurun_out/bench_trace_w8_$code
rm -rf $OUT; mkdir -p $OUT
( cd $ROOTDIR && RGCN_LIBRARY=devtools RGCN_GEMM_W8=$code timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python bench.py --steps 30 --warmup 10 --cpu-steps 0 --no-kernel-profile --no-extra-workloads --no-fp32-reference --no-live-traffic --gemm-mode 6 ) > $OUT/log.txt 2>&1
echo code $code rc=$?
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $ROOTDIR/tools/train_step_timeline.py $f k_colsum_final > $ROOTDIR/gpurun_out/timeline_w8_$code.txt 2>&1
rm -rf $OUT
done
