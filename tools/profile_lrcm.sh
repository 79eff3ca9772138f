#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
N_POINTS=1000000 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_lrcm -- python3 $R/tools/lrcm_1e6.py > $R/gpurun_out/prof_lrcm.log 2>&1
f=$(find $R/gpurun_out/prof_lrcm -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_summary.py $f $R/gpurun_out/prof_lrcm_summary.json
rm -rf $R/gpurun_out/prof_lrcm
