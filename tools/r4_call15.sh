#!/bin/bash
# persistent mode of the contraction kernel: parity (whole suite touches it), then A/B of the products and of the headline
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q -k "not multirank and not bench_gpus and not full_size and not c2_parity" > gpurun_out/r4_t15.log 2>&1; echo rc=$? >> gpurun_out/r4_t15.log; tail -4 gpurun_out/r4_t15.log
grep -q "rc=0" gpurun_out/r4_t15.log || exit 1
for v in 1 0; do
  echo "== GSI_GEMM_PERSIST=$v"
  GSI_GEMM_PERSIST=$v timeout -k 10 120 python tools/bench_lrcm_products.py --samples 1024 --l 320 --reps 4
  GSI_GEMM_PERSIST=$v timeout -k 10 120 python tools/bench_gemm.py --grid 256 --l 160
  GSI_GEMM_PERSIST=$v python bench.py --steps 10 --no-secondary --no-full-parity --no-cpu-baseline > gpurun_out/r4_bench_p$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4_bench_p$v.json"))
print(round(d["ms_per_step"],2), round(d["value"],1), round(d["roofline"]["frac"],3), {k:round(v,2) for k,v in d["phases_ms_per_step"].items()})
PY
done
