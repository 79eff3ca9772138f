#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_impl_$c -- python3 $R/tools/implicit_product.py 1000 320 > $R/gpurun_out/pmc_impl_$c.log 2>&1 || exit 1
  tail -1 $R/gpurun_out/pmc_impl_$c.log
done
python3 - <<PY
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$R/gpurun_out/pmc_impl_%s/*/*counter_collection.csv"%c)[0]
    for r in csv.DictReader(open(f)):
        if "gemm_f64_kernel" in r["Kernel_Name"] and r["Counter_Name"]==c:
            print(c, r["Kernel_Name"][:60], float(r["Counter_Value"])/1e6*(2 if c=="FETCH_SIZE" else 1), "GB (KB*1024, fetch doubled)", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms")
PY
rm -rf $R/gpurun_out/pmc_impl_FETCH_SIZE $R/gpurun_out/pmc_impl_WRITE_SIZE
