#!/bin/bash
# PMC counters of the pre-split-weight GEMM kernels; counters only, no traces
export TMPDIR=/tmp
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/w8_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p1 -o p1 -- python tools/gemm_w8_pmc.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/p2 -o p2 -- python tools/gemm_w8_pmc.py > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES -d $OUT/p3 -o p3 -- python tools/gemm_w8_pmc.py > $OUT/p3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_REQ_sum -d $OUT/p5 -o p5 -- python tools/gemm_w8_pmc.py > $OUT/p5.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python tools/gemm_w8_pmc.py > $OUT/kt.log 2>&1
tail -3 $OUT/p1.log | cut -c1-200
for p in p1 p2 p3 p5; do python tools/pmc_summary.py $OUT/$p/${p}_results.db k_gemm; done 2>&1 | tee gpurun_out/w8_pmc_$TAG.txt
grep -h "k_gemm" $OUT/kt/*kernel_stats.csv 2>/dev/null | cut -c1-220 | tee -a gpurun_out/w8_pmc_$TAG.txt
