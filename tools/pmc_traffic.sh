#!/bin/bash
# HBM traffic of the operator passes: two separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass),
# --kernel-trace only, on tools/bench_gemm.py (C2 shapes).  usage (GPU box): bash tools/pmc_traffic.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_$c -- python3 $R/tools/bench_gemm.py --reps 2 > $R/gpurun_out/pmc_traffic_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_traffic_summary.py $R/gpurun_out $R/gpurun_out/gemm_traffic.json
