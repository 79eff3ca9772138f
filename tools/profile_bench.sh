#!/bin/bash
# usage (on the GPU box): bash tools/profile_bench.sh [tag] [extra bench.py flags]
#   -> gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats of the bench command), per-dispatch summary JSON,
#      kernel stats CSV.  The program itself follows `--` (no env/bash hop: the profiler has initialised the GPU).
R=$GRAFT_REPO_ROOT
TAG=${1:-bench}
shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-boundary "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_stats.csv \;
T=$(find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_summary.py $T $R/gpurun_out/prof_${TAG}_dispatch_summary.json | head -40
tail -c 300 $R/gpurun_out/prof_$TAG.log
