#!/bin/bash
# where do the LowRankCovMatrix contractions lose against the stored square operand?  time per product against N_s (NN: K = N_s,
# per-workgroup overhead is the intercept) and against l (one 160-column chunk vs two)
R=$GRAFT_REPO_ROOT; cd $R
for ns in 512 1024 2048 4096; do timeout -k 10 120 python tools/bench_lrcm_products.py --samples $ns --l 320 --reps 4; done
for l in 160 320; do timeout -k 10 120 python tools/bench_lrcm_products.py --samples 1024 --l $l --reps 4; done
timeout -k 10 120 python tools/bench_gemm.py --grid 256 --l 160
timeout -k 10 120 python tools/bench_gemm.py --grid 256 --l 320
python -m pytest tests -m gpu -x -q -k "svd or randsvd or golden" 2>&1 | tail -2
python bench.py --steps 10 --no-secondary --no-full-parity --no-cpu-baseline > gpurun_out/r4_bench_z.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_z.json"))
print(round(d["ms_per_step"],2), round(d["value"],1), {k:round(v,2) for k,v in d["phases_ms_per_step"].items()})
PY
