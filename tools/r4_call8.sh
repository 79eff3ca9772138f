#!/bin/bash
# round-4 GPU call 8: generated-operand kernels after the VALU diet (table: v_sad_u32 offsets; points: interior path,
# rsq-based sqrt, clamped exp): parity tests, then the benches
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q -k "implicit or pointcov or gridcov" > gpurun_out/r4_t8.log 2>&1; echo rc=$? >> gpurun_out/r4_t8.log; tail -4 gpurun_out/r4_t8.log
grep -q "rc=0" gpurun_out/r4_t8.log || exit 1
{
timeout -k 10 300 python tools/pointcov_bench.py || exit 1
timeout -k 10 300 python tools/implicit_ab.py 500 1 || exit 1
timeout -k 10 300 python tools/implicit_ab.py 500 0 || exit 1
} > gpurun_out/r04_generated_operand_bench.log 2>&1
cat gpurun_out/r04_generated_operand_bench.log
