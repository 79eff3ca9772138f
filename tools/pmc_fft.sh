#!/bin/bash
# PMC passes over the FFT covariance product (tools/fft_cov_bench.py --no-svd): bash tools/pmc_fft.sh <tag> [bench flags]
#   -> gpurun_out/pmc_fft_<tag>.json : per fft_pass_kernel<MODE> averages (duration, FETCH_SIZE x2, WRITE_SIZE, SQ / LDS counters)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fft_${TAG}_$i -- python3 $R/tools/fft_cov_bench.py --no-svd "$@" > $R/gpurun_out/pmc_fft_${TAG}_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_fft_summary.py $R/gpurun_out $TAG
