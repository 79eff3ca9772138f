#!/bin/bash
# final: whole -m gpu suite + smoke, then the profile set of tools/r4_call12.sh
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t16.log 2>&1; echo rc=$? >> gpurun_out/r4_t16.log; tail -5 gpurun_out/r4_t16.log
grep -q "rc=0" gpurun_out/r4_t16.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()"
bash tools/r4_call12.sh
