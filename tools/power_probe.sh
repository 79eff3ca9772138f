#!/bin/bash
# usage: tools/power_probe.sh <ablate> : runs the GEMM bench ~6 s and samples socket power / sclk
A=$1
(GSI_GEMM_ABLATE=$A timeout -k 5 60 python tools/bench_gemm.py --reps 120 > gpurun_out/pp_$A.log 2>&1 &)
sleep 4.0
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -E 's/.*\((.*Mhz)\).*/\1/; s/.*Power \(W\): //' | tr "\n" " "; sleep 0.5; done
echo
sleep 5
tail -2 gpurun_out/pp_$A.log
