#!/bin/bash
# PMC counters of the decoder's entity-gradient kernel (k_dec_entity_lines / k_dec_entity_grad) inside the device train
# step: L2 hit rate, fabric bytes, wave occupancy.  Counters only, separate passes (MI355X_MICROARCH.md).
# Usage: tools/gpu_dec_pmc.sh TAG [ENV=VAL ...]
export TMPDIR=/tmp
TAG=${1:-dec}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/dec_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
CMD="python tools/train_step_trace.py 6"
export RGCN_STREAMS=0 "$@"
pass() { n=$1; shift; timeout 120 rocprofv3 --pmc "$@" -d $OUT/$n -o $n -- $CMD > $OUT/$n.log 2>&1; python tools/pmc_summary.py $OUT/$n/${n}_results.db "k_dec_ent" | tee -a $OUT/summary.txt; }
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES
pass p2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pass p4 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass p6 FETCH_SIZE
rm -rf $OUT/p*/
