#!/bin/bash
# rocprofv3 passes over the device train step (tools/train_step_trace.py: rgcn_train_step_device on resident inputs, N = 330,000
# decoder triples): kernel trace + FETCH_SIZE / WRITE_SIZE, summary -> gpurun_out/<tag>_rocprof_train_step[_fb15k].md
#   tools/gpu_profile_train.sh TAG [fb15k]
TAG=${1:-r02}; WL=$2
export TMPDIR=/tmp
export RGCN_STREAMS=0
CMD="python tools/train_step_trace.py 12 $WL"
ROOTDIR=$GRAFT_REPO_ROOT
SUF=""; [ -n "$WL" ] && SUF="_$WL"
OUT=$ROOTDIR/gpurun_out/prof_${TAG}_train_step${SUF}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
( cd $ROOTDIR && timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD ) > $OUT/trace.log 2>&1; echo "trace rc=$?"
( cd $ROOTDIR && timeout 180 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD ) > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
( cd $ROOTDIR && timeout 180 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD ) > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
cd $ROOTDIR
python tools/rocprof_summary.py $OUT gpurun_out/${TAG}_rocprof_train_step${SUF}.md "RGCN_STREAMS=0 $CMD" | head -45
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
