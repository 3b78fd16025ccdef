#!/bin/bash
# kernel trace of the first 120 steps of a fresh process (tools/bench_ramp.py), one line per step (tools/step_series.py)
export TMPDIR=/tmp
ROOTDIR=$GRAFT_REPO_ROOT
OUT=$ROOTDIR/gpurun_out/ramp_trace
mkdir -p $OUT
cd /tmp
( cd $ROOTDIR && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python tools/bench_ramp.py ) > $OUT/log.txt 2>&1
echo rc=$?
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $ROOTDIR/tools/step_series.py $f > $OUT/series.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete
head -70 $OUT/series.txt; tail -8 $OUT/series.txt
