#!/bin/bash
# PMC counters of the GEMM kernels (RGCN_GEMM_MODE picks the arithmetic); counters only, no traces.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p1 -o p1 -- python tools/gemm_pmc.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/p2 -o p2 -- python tools/gemm_pmc.py > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES -d $OUT/p3 -o p3 -- python tools/gemm_pmc.py > $OUT/p3.log 2>&1
tail -3 $OUT/p1.log | cut -c1-200; tail -3 $OUT/p3.log | cut -c1-200
python tools/pmc_summary.py $OUT/p1/p1_results.db k_gemm
python tools/pmc_summary.py $OUT/p2/p2_results.db k_gemm
python tools/pmc_summary.py $OUT/p3/p3_results.db k_gemm
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/p4 -o p4 -- python tools/gemm_pmc.py > $OUT/p4.log 2>&1
python tools/pmc_summary.py $OUT/p4/p4_results.db k_gemm
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum -d $OUT/p5 -o p5 -- python tools/gemm_pmc.py > $OUT/p5.log 2>&1
python tools/pmc_summary.py $OUT/p5/p5_results.db k_gemm
tail -2 $OUT/p4.log $OUT/p5.log | cut -c1-300
