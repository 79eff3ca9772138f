#!/bin/bash
# the round-4 profile set -- the driver's command, its kernel trace, its PMC traffic passes; FFT kernel stats + PMC
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 700 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo bench_rc=$?
bash tools/profile_bench.sh r04 > /dev/null 2>&1 || exit 1
cp gpurun_out/prof_r04_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv; cp gpurun_out/prof_r04_dispatch_summary.json gpurun_out/r04_bench_dispatch_summary.json
bash tools/pmc_bench_traffic.sh > gpurun_out/r04_pmc_bench_traffic.log 2>&1; tail -3 gpurun_out/r04_pmc_bench_traffic.log
rm -rf gpurun_out/prof_r04 gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
bash tools/fft_profile.sh r04 > /dev/null 2>&1
bash tools/pmc_fft.sh r04_1024sq --Ns 1024 1024 --l 256 --fftrf > gpurun_out/r04_pmc_fft_1024sq.log 2>&1 || echo pmc1024_failed
bash tools/pmc_fft.sh r04_512cube --Ns 512 512 512 --l 16 --fftrf > gpurun_out/r04_pmc_fft_512cube.log 2>&1 || echo pmc512_failed
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fft512_r04 -- python3 $R/tools/fft_cov_bench.py --Ns 512 512 512 --l 16 --fftrf --no-svd > $R/gpurun_out/prof_fft512_r04.log 2>&1
find $R/gpurun_out/prof_fft512_r04 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04_fft_kernel_stats_512cube.csv \;
cd $R
# the scattered-point operator: default (96 x 320 tiles), the A/B knobs, the kernel trace
(echo "# tools/pointcov_bench.py (default: 96 x 320 tiles over the packed sketch panel, K splits chosen outright)"; timeout -k 10 200 python tools/pointcov_bench.py
 echo "# GSI_POINTCOV_ROWS=64 (64 x 320 tiles)"; GSI_POINTCOV_ROWS=64 timeout -k 10 200 python tools/pointcov_bench.py
 echo "# GSI_GEMM_FORCE_SPLIT=1 / 2 (96 rows: the partial last round of workgroups)"
 GSI_GEMM_FORCE_SPLIT=1 timeout -k 10 200 python tools/pointcov_bench.py | grep exponential; GSI_GEMM_FORCE_SPLIT=2 timeout -k 10 200 python tools/pointcov_bench.py | grep exponential
 echo "# GSI_POINTCOV_WIDE=0 (128 x 160 tiles, every entry generated twice at l = 320)"; GSI_POINTCOV_WIDE=0 timeout -k 10 200 python tools/pointcov_bench.py
 echo "# l = 160 (one column chunk: the 128 x 160 kernel)"; timeout -k 10 200 python tools/pointcov_bench.py 450 160
) > gpurun_out/r04_pointcov_bench.log 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pc -- python3 $R/tools/pointcov_bench.py > $R/gpurun_out/prof_pc.log 2>&1
find $R/gpurun_out/prof_pc -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04_pointcov_kernel_stats.csv \;
cd $R
rm -rf gpurun_out/prof_pc
rm -rf gpurun_out/prof_fft_r04 gpurun_out/prof_fft3_r04 gpurun_out/prof_fft512_r04 gpurun_out/pmc_fft_r04_*_[1-4]
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_bench_line.json"))
print(round(d["ms_per_step"],2), round(d["value"],1), round(d["roofline"]["frac"],3), {k:round(v,2) for k,v in d["phases_ms_per_step"].items()})
for k,v in d["secondary"].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,str))})
PY
