#!/bin/bash
# FFT covariance operator: timings for DESIGN 4.6 + a rocprofv3 kernel-trace summary.  usage (GPU box): bash tools/fft_profile.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
cd $R
{
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 1000 1000 --l 256
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 1024 1024 --l 256 --fftrf
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 256 256 --l 160
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 128 128 128 --l 64
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 256 256 256 --l 64
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 2048 2048 --l 64 --no-svd
timeout -k 10 200 python tools/fft_cov_bench.py --Ns 4096 --l 64 --no-svd
} > gpurun_out/${TAG}_fft_cov_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fft_$TAG -- python3 $R/tools/fft_cov_bench.py --Ns 1024 1024 --l 256 --fftrf --no-svd > $R/gpurun_out/prof_fft_$TAG.log 2>&1
find $R/gpurun_out/prof_fft_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_fft_kernel_stats_1024sq.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fft3_$TAG -- python3 $R/tools/fft_cov_bench.py --Ns 128 128 128 --l 64 --no-svd > $R/gpurun_out/prof_fft3_$TAG.log 2>&1
find $R/gpurun_out/prof_fft3_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_fft_kernel_stats_128cube.csv \;
cat $R/gpurun_out/${TAG}_fft_cov_bench.log
