#!/bin/bash
# round-4 GPU call 6a: the driver's command (full line), its rocprofv3 kernel trace + stats, its two PMC traffic passes
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 700 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo bench_rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_bench_line.json"))
print(round(d["ms_per_step"],2), round(d["value"],1), round(d["roofline"]["frac"],3), {k:round(v,2) for k,v in d["phases_ms_per_step"].items()})
for k,v in d["secondary"].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,str))})
print(d["sv_rel_err"], d["xis_err_up_to_sign"], d["cpu_baseline"]["value"])
PY
bash tools/profile_bench.sh r04 || exit 1
cp gpurun_out/prof_r04_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv; cp gpurun_out/prof_r04_dispatch_summary.json gpurun_out/r04_bench_dispatch_summary.json
bash tools/pmc_bench_traffic.sh > gpurun_out/r04_pmc_bench_traffic.log 2>&1; tail -5 gpurun_out/r04_pmc_bench_traffic.log
rm -rf gpurun_out/prof_r04 gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
