#!/bin/bash
# A/B of the intermediate layout of the FFT covariance passes (GSI_FFT_TB: 0 natural, 8 / 16 blocked): product times only.
# usage (GPU box): bash tools/fft_layout_ab.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}
cd $R
for TB in 0 8 16; do
  echo "=== GSI_FFT_TB=$TB"
  export GSI_FFT_TB=$TB
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 1000 1000 --l 256 --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 1024 1024 --l 256 --fftrf --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 2048 2048 --l 64 --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 128 128 128 --l 64 --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 256 256 256 --l 64 --no-svd || exit 1
  timeout -k 10 300 python tools/fft_cov_bench.py --Ns 512 512 512 --l 16 --fftrf --no-svd || exit 1
done > gpurun_out/${TAG}_fft_layout_ab.log 2>&1
cat gpurun_out/${TAG}_fft_layout_ab.log
