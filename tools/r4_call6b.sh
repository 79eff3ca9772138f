#!/bin/bash
# round-4 GPU call 6b: FFT operator profiles (kernel stats + PMC, 1024^2 and 512^3), multi-rank rehearsal with the
# degeneracy-robust figures, C5 against the oracle (printed numbers)
R=$GRAFT_REPO_ROOT; cd $R
bash tools/fft_profile.sh r04 > /dev/null 2>&1; cat gpurun_out/r04_fft_cov_bench.log
bash tools/pmc_fft.sh r04_1024sq --Ns 1024 1024 --l 256 --fftrf > gpurun_out/r04_pmc_fft_1024sq.log 2>&1 || echo pmc1024_failed
bash tools/pmc_fft.sh r04_512cube --Ns 512 512 512 --l 16 --fftrf > gpurun_out/r04_pmc_fft_512cube.log 2>&1 || echo pmc512_failed
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fft512_r04 -- python3 $R/tools/fft_cov_bench.py --Ns 512 512 512 --l 16 --fftrf --no-svd > $R/gpurun_out/prof_fft512_r04.log 2>&1
find $R/gpurun_out/prof_fft512_r04 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04_fft_kernel_stats_512cube.csv \;
cd $R
rm -rf gpurun_out/prof_fft_r04 gpurun_out/prof_fft3_r04 gpurun_out/prof_fft512_r04 gpurun_out/pmc_fft_r04_*_[1-4]
timeout -k 10 500 python tools/multirank_rehearsal.py --worlds 2 4 --case fft3d fft2d > gpurun_out/r04_multirank_rehearsal_one_gpu.jsonl 2> gpurun_out/r04_multirank_rehearsal.err
timeout -k 10 300 python tools/multirank_rehearsal.py --worlds 2 --case fft3d_asym fft2d_asym >> gpurun_out/r04_multirank_rehearsal_one_gpu.jsonl 2>> gpurun_out/r04_multirank_rehearsal.err
cut -c1-420 gpurun_out/r04_multirank_rehearsal_one_gpu.jsonl
python -m pytest tests -m gpu -x -q -s -k "c5" 2>&1 | grep -i "C5 at\|passed\|failed" 
