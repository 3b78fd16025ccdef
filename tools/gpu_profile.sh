#!/bin/bash
# rocprofv3 passes over the default bench command: kernel trace + stats, then PMC passes for HBM bytes
# (FETCH_SIZE and WRITE_SIZE in SEPARATE runs: they do not fit one pass on gfx950 -- MI355X_MICROARCH.md).
TAG=${1:-r01}
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
CMD="python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-kernel-profile"
# second argument "serial": exclusive kernel durations (no side streams, no pipelined prep)
if [ "$2" = "serial" ]; then export RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0; fi
cd /tmp
ROOTDIR=$GRAFT_REPO_ROOT
OUT=$ROOTDIR/gpurun_out/prof_$TAG
( cd $ROOTDIR && rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD ) > $OUT/trace.log 2>&1
echo "trace rc=$?"
( cd $ROOTDIR && rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD ) > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
( cd $ROOTDIR && rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD ) > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
find $OUT -type f | head -50
ls -la $OUT/*/* | head
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -40 $f; done
for f in $(find $OUT -name "*counter_collection.csv" | head -2); do echo "== $f"; head -5 $f; wc -l $f; done
# keep the merge small: drop the big per-dispatch traces, keep stats + counters
find $OUT -name "*kernel_trace.csv" -size +20M -delete
