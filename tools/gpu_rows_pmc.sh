#!/bin/bash
# PMC counters of the destination-major banded block layer kernel (k_block_rows): where do its cycles go?  Counters
# only, separate passes (MI355X_MICROARCH.md).  Usage: tools/gpu_rows_pmc.sh TAG [workload] [ENV=VAL ...]
export TMPDIR=/tmp
TAG=${1:-rows}; shift
WL=${1:-fb237_block}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/rows_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
CMD="python bench.py --workload $WL --steps 6 --warmup 2 --cpu-steps 0 --no-kernel-profile --no-extra-workloads --no-fp32-reference --no-live-traffic"
export RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0 "$@"
pass() { n=$1; shift; timeout 120 rocprofv3 --pmc "$@" -d $OUT/$n -o $n -- $CMD > $OUT/$n.log 2>&1; python tools/pmc_summary.py $OUT/$n/${n}_results.db "k_block_rows" | tee -a $OUT/summary.txt; }
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass p2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES
pass p3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS
pass p4 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum
pass p6 FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
rm -rf $OUT/p*/  # keep the logs / printed summaries only
