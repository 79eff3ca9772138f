#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q -k "fft" > gpurun_out/r4_t7.log 2>&1; echo rc=$? >> gpurun_out/r4_t7.log; tail -3 gpurun_out/r4_t7.log
grep -q "rc=0" gpurun_out/r4_t7.log || exit 1
for TB in 0 16 32 64; do
  echo "=== GSI_FFT_TB=$TB"
  export GSI_FFT_TB=$TB
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 1000 1000 --l 256 --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 2048 2048 --l 64 --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 256 256 256 --l 64 --no-svd || exit 1
  timeout -k 10 300 python tools/fft_cov_bench.py --Ns 512 512 512 --l 16 --fftrf --no-svd || exit 1
done > gpurun_out/r04_fft_layout_ab2.log 2>&1
cat gpurun_out/r04_fft_layout_ab2.log
unset GSI_FFT_TB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fftb -- python3 $R/tools/fft_cov_bench.py --Ns 1024 1024 --l 256 --fftrf --no-svd > $R/gpurun_out/prof_fftb.log 2>&1
find $R/gpurun_out/prof_fftb -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04b_fft_kernel_stats_1024sq.csv \;
rm -rf $R/gpurun_out/prof_fftb
grep fft_pass $R/gpurun_out/r04b_fft_kernel_stats_1024sq.csv | cut -d, -f1-4 | cut -c1-60,200-
