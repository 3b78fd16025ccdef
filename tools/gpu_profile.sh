#!/bin/bash
# rocprofv3 passes over one bench workload: kernel trace + stats, then PMC passes for HBM bytes (FETCH_SIZE and
# WRITE_SIZE in SEPARATE runs: they do not fit one pass on gfx950 -- MI355X_MICROARCH.md); summaries go to
# gpurun_out/ as <tag>_rocprof_<workload>[_serial].md (+ _traffic.json): copy the ones to keep into profiles/.
#   tools/gpu_profile.sh TAG WORKLOAD [serial]     serial: exclusive kernel durations (no side streams, no pipelined prep)
TAG=${1:-r02}; WL=${2:-fb237_block}; MODE=$3
export TMPDIR=/tmp
SUF=""; if [ "$MODE" = "serial" ]; then export RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0; SUF="_serial"; fi
CMD="python bench.py --workload $WL --steps 20 --warmup 5 --cpu-steps 0 --no-kernel-profile --no-extra-workloads --no-fp32-reference"
ROOTDIR=$GRAFT_REPO_ROOT
OUT=$ROOTDIR/gpurun_out/prof_${TAG}_${WL}${SUF}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
( cd $ROOTDIR && timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD ) > $OUT/trace.log 2>&1; echo "trace rc=$?"
( cd $ROOTDIR && timeout 180 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD ) > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
( cd $ROOTDIR && timeout 180 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD ) > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
cd $ROOTDIR
python tools/rocprof_summary.py $OUT gpurun_out/${TAG}_rocprof_${WL}${SUF}.md "$([ "$MODE" = serial ] && echo 'RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0 ')$CMD" | head -40
# keep the merge small: drop the per-dispatch databases, the summaries are what is kept
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
