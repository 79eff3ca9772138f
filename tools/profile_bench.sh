#!/bin/bash
# usage (on the GPU box): bash tools/profile_bench.sh   -> gpurun_out/prof_bench/*kernel_stats.csv + bench line
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_bench_kernel_stats.csv \;
tail -c 600 $R/gpurun_out/prof_bench.log
