#!/bin/bash
# HBM traffic of the bench command itself: two separate rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE do not fit one
# pass) with --kernel-trace only, on `bench.py` (headline workload, short).  usage (GPU box): bash tools/pmc_bench_traffic.sh
#   -> gpurun_out/pmc_bench_{FETCH_SIZE,WRITE_SIZE}/, gpurun_out/r05_bench_traffic.json (copy to profiles/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bench_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-boundary > $R/gpurun_out/pmc_bench_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_bench_summary.py $R/gpurun_out $R/gpurun_out/r05_bench_traffic.json
