#!/bin/bash
export TMPDIR=/tmp
ROOTDIR=$GRAFT_REPO_ROOT
OUT=$ROOTDIR/gpurun_out/train_trace
mkdir -p $OUT
cd /tmp
( cd $ROOTDIR && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python tools/train_step_trace.py 12 ) > $OUT/log.txt 2>&1
echo rc=$?
tail -3 $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $ROOTDIR/tools/train_step_timeline.py $f > $OUT/timeline.txt 2>&1
head -5 $f | cut -c1-400
find $OUT -name "*kernel_trace.csv" -size +20M -delete
cat $OUT/timeline.txt | head -100
