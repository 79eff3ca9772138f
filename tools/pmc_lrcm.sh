#!/bin/bash
# SQ counter passes over the two LowRankCovMatrix contractions: bash tools/pmc_lrcm.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 $R/tools/bench_lrcm_products.py --reps 2 > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 $R/tools/bench_lrcm_products.py --reps 2 > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -- python3 $R/tools/bench_lrcm_products.py --reps 2 > $R/gpurun_out/pmc_${TAG}_c.log 2>&1
cd $R && python3 tools/pmc_summary.py $TAG gpurun_out/pmc_${TAG}_summary.json
