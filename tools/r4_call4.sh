#!/bin/bash
# round-4 GPU call 4: the whole -m gpu suite on the deferred thin-Q / in-loader pointcov / blocked FFT build, the fp64
# VALU-vs-MFMA microbenchmark, a short headline bench
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t4.log 2>&1; echo rc=$? >> gpurun_out/r4_t4.log; tail -6 gpurun_out/r4_t4.log
timeout -k 10 120 ./tools/mfma_valu_f64_conflict > gpurun_out/r04_mfma_valu_f64_conflict.log 2>&1; cat gpurun_out/r04_mfma_valu_f64_conflict.log
timeout -k 10 600 python bench.py --steps 10 --no-secondary --no-full-parity > gpurun_out/r4_bench_short.json 2> gpurun_out/r4_bench_short.err; echo bench_rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_short.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["phases_ms_per_step"], d["path_counters"], d["sv_rel_err"], d["xis_err_up_to_sign"])
PY
