for cfg in "64 152" "32 152" "32 76" "20 40" "40 40" "64 76"; do
  set -- $cfg
  echo "B0=$1 B1=$2"
  GSI_FFT_B0=$1 GSI_FFT_B1=$2 timeout -k 10 300 python tools/fft_cov_bench.py --Ns 128 128 128 --l 64 --no-svd | cut -c1-110
  GSI_FFT_B0=$1 GSI_FFT_B1=$2 timeout -k 10 300 python tools/fft_cov_bench.py --Ns 1000 1000 --l 128 --no-svd | cut -c1-110
  GSI_FFT_B0=$1 GSI_FFT_B1=$2 timeout -k 10 300 python tools/fft_cov_bench.py --Ns 250 250 --l 160 --no-svd | cut -c1-110
done
