for cfg in "128 1" "64 1" "64 2" "128 2"; do set -- $cfg; GSI_LU_RK_CHUNK=$1 GSI_LU_RK_DEPTH=$2 python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $1 depth $2: step', round(d['ms_per_step'],2), 'lu', round(d['phases_ms_per_step']['lu'],2), 'gemm', round(d['phases_ms_per_step']['gemm_n']+d['phases_ms_per_step']['gemm_t'],1))"; done
