#!/bin/bash
# round-4 GPU call 5: lean-VALU Gram / triangular-product kernels: parity tests, A/B of the headline QR phase
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q -k "qr or svd or randsvd or golden or c2_parity or rangefinder or nystrom" > gpurun_out/r4_t5.log 2>&1; echo rc=$? >> gpurun_out/r4_t5.log; tail -4 gpurun_out/r4_t5.log
grep -q "rc=0" gpurun_out/r4_t5.log || exit 1
for v in "fast" "GSI_SY_NO_FAST=1 GSI_TR_NO_FAST=1"; do
  echo "== $v"
  env $( [ "$v" = fast ] || echo $v ) python bench.py --steps 10 --no-secondary --no-full-parity --no-cpu-baseline > gpurun_out/r4_bench_ab.json 2> gpurun_out/r4_bench_ab.err || exit 1
  python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench_ab.json"))
print(round(d["ms_per_step"],2), round(d["value"],1), round(d["roofline"]["frac"],3), {k:round(v,2) for k,v in d["phases_ms_per_step"].items()}, d["phases_hbm"]["qr"]["ms_per_factorization"])
PY
done
