#!/bin/bash
# PMC counters of the HBM-bound kernels at FULL-GRAPH scale (272,115 edges): what holds k_combine at 3.5 TB/s there?
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fullgraph_pmc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD="python bench.py --workload fb237_block_traingraph --steps 4 --warmup 2 --cpu-steps 0 --no-kernel-profile --no-extra-workloads --no-fp32-reference"
export RGCN_STREAMS=0 RGCN_BENCH_PREFETCH=0
pass() { n=$1; shift; timeout 90 rocprofv3 --pmc "$@" -d $OUT/$n -o $n -- $CMD > $OUT/$n.log 2>&1; for k in k_block_msg_fwd k_combine; do python tools/pmc_summary.py $OUT/$n/${n}_results.db $k; done; }
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass p2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES
pass p4 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass p5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum
rm -rf $OUT/p*/
