#!/bin/bash
# kernel timeline of the default bench step (pipelined): rocprofv3 --kernel-trace, then tools/train_step_timeline.py
export TMPDIR=/tmp
ROOTDIR=$GRAFT_REPO_ROOT
OUT=$ROOTDIR/gpurun_out/bench_trace
mkdir -p $OUT
cd /tmp
( cd $ROOTDIR && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-kernel-profile --no-extra-workloads --no-fp32-reference --gemm-mode 6 ) > $OUT/log.txt 2>&1
echo rc=$?
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $ROOTDIR/tools/train_step_timeline.py $f k_colsum_final > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt | head -120
