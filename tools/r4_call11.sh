#!/bin/bash
# round-4 GPU call 11: the whole -m gpu suite on the final build
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t11.log 2>&1; echo rc=$? >> gpurun_out/r4_t11.log; tail -6 gpurun_out/r4_t11.log
python -c "import __graft_entry__ as g; g.smoke()"
