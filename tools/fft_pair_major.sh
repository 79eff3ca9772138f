#!/bin/bash
# column pairs "pair-major" -- batches small enough that the intermediate between the passes (N1 x M0 complex
# per pair at 2-D: 32.8 MB at 1000^2) stays in the 256 MB Infinity Cache -- against the default 2 GB batches.
cd $GRAFT_REPO_ROOT
for mb in 2048 1024 512 256 200 128 64; do
  echo "# GSI_FFT_W_MB=$mb"
  GSI_FFT_W_MB=$mb timeout -k 10 100 python tools/fft_cov_bench.py --Ns 1000 1000 --l 256 --no-svd | cut -c1-150
  GSI_FFT_W_MB=$mb timeout -k 10 100 python tools/fft_cov_bench.py --Ns 1024 1024 --l 256 --fftrf --no-svd | cut -c1-150
  GSI_FFT_W_MB=$mb timeout -k 10 100 python tools/fft_cov_bench.py --Ns 500 500 --l 256 --no-svd | cut -c1-150
done
