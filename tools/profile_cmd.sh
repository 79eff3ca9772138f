#!/bin/bash
# usage (on the GPU box): bash tools/profile_cmd.sh <tag> <python script and args...>  -> per-kernel dispatch summary
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_summary.py $f $R/gpurun_out/prof_${TAG}_summary.json
rm -rf $R/gpurun_out/prof_$TAG
