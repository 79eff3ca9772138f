#!/bin/bash
# round-4 GPU call 3: pointcov in-loader generator (tests + bench A/B), FFT tile-size A/B on the blocked layout, the N > 1 rehearsal
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q -k "pointcov or fft_powerlaw_operator or c5" > gpurun_out/r4_t3.log 2>&1; echo rc=$? >> gpurun_out/r4_t3.log; tail -4 gpurun_out/r4_t3.log
grep -q "rc=0" gpurun_out/r4_t3.log || exit 1
{
echo "== in-loader generator"; timeout -k 10 300 python tools/pointcov_bench.py || exit 1
echo "== round-3 row panels (GSI_POINTCOV_PANELS=1)"; GSI_POINTCOV_PANELS=1 timeout -k 10 300 python tools/pointcov_bench.py || exit 1
} > gpurun_out/r04_pointcov_bench.log 2>&1
cat gpurun_out/r04_pointcov_bench.log
{
for cfg in "16 4 140" "16 2 140" "16 2 70" "16 1 40"; do
  set -- $cfg
  echo "=== GSI_FFT_TB=$1 GSI_FFT_MIN_T=$2 GSI_FFT_B1=$3"
  export GSI_FFT_TB=$1 GSI_FFT_MIN_T=$2 GSI_FFT_B1=$3
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 1000 1000 --l 256 --no-svd || exit 1
  timeout -k 10 200 python tools/fft_cov_bench.py --Ns 256 256 256 --l 64 --no-svd || exit 1
  timeout -k 10 300 python tools/fft_cov_bench.py --Ns 512 512 512 --l 16 --fftrf --no-svd || exit 1
done
unset GSI_FFT_TB GSI_FFT_MIN_T GSI_FFT_B1
} > gpurun_out/r04_fft_tile_ab.log 2>&1
cat gpurun_out/r04_fft_tile_ab.log
GSI_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 3 > gpurun_out/r4_rehearse_2.json 2> gpurun_out/r4_rehearse_2.err; echo bench_rc=$?
