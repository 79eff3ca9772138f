#!/bin/bash
# SQ counters of the scattered-point contraction (pointcov_wide_kernel) and the table-generated one beside it:
#   bash tools/pmc_pointcov.sh  ->  gpurun_out/r04_pointcov_sq_counters.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pc -- python3 $R/tools/pointcov_bench.py > $R/gpurun_out/pmc_pc.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$R/gpurun_out/pmc_pc/*/*counter_collection.csv")[0]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = "pointcov_wide_kernel" if "pointcov_wide_kernel" in r["Kernel_Name"] else ("gemm_f64_kernel<10, false, 1, 0>" if "gemm_f64_kernel<10, false, 1, 0>" in r["Kernel_Name"] else None)
    if k is None: continue
    out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[k]["duration_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {}
for k, v in out.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    mf = a["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0   # MOPS counts 4 per v_mfma_f64_16x16x4 wave instruction... kept raw as well
    a["valu_other_than_mfma_per_mfma"] = (a["SQ_INSTS_VALU"] - a["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0) / (a["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0)
    a["clock_GHz"] = a["GRBM_GUI_ACTIVE"] / 8 / (a["duration_ms"] * 1e6)
    a["mfma_busy_frac"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8)
    res[k] = {c: (round(x, 4) if abs(x) < 1e4 else float(f"{x:.5g}")) for c, x in a.items()}
json.dump(res, open("$R/gpurun_out/r04_pointcov_sq_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $R/gpurun_out/pmc_pc
