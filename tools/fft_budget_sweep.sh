cd $GRAFT_REPO_ROOT
for b0 in 32 64 128; do for b1 in 40 76 152; do
echo "B0=$b0 B1=$b1"
GSI_FFT_B0=$b0 GSI_FFT_B1=$b1 timeout -k 10 100 python tools/fft_cov_bench.py --Ns 1024 1024 --l 256 --fftrf --no-svd | cut -c1-110
GSI_FFT_B0=$b0 GSI_FFT_B1=$b1 timeout -k 10 100 python tools/fft_cov_bench.py --Ns 128 128 128 --l 64 --no-svd | cut -c1-110
done; done
echo 512cube
timeout -k 10 300 python tools/fft_cov_bench.py --Ns 512 512 512 --l 16 --q 2
