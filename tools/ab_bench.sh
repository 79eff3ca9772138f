#!/bin/bash
# A/B the whole bench step on one box: shipped library vs $1, alternating, 2 rounds
for r in 1 2; do
  for lib in "" "$1"; do
    GSI_HIP_LIB=$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('${lib:-shipped}', round(d['ms_per_step'],2), round(p['gemm_n'],2), round(p['gemm_t'],2))" || exit 1
  done
done
