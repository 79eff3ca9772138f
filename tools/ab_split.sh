for leg in 0 1; do
  if [ $leg = 1 ]; then export GSI_GEMM_SPLIT_LEGACY=1; fi
  echo "legacy=$leg"
  timeout -k 10 200 python tools/bench_gemm.py --grid 224 --reps 5 || exit 1
  timeout -k 10 300 python tools/bench_gemm_shard.py --n 92672 --G 2 --reps 5 || exit 1
  timeout -k 10 300 python tools/bench_gemm_shard.py --n 185344 --G 8 --reps 5 || exit 1
done
