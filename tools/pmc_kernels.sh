#!/bin/bash
# SQ counters of named kernels inside the bench command: bash tools/pmc_kernels.sh <tag> <kernel substring> [...]
#   -> gpurun_out/pmc_k_<tag>.json (per kernel averages)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k_${TAG}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_k_${TAG}_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_kernel_summary.py $R/gpurun_out $TAG "$@"
