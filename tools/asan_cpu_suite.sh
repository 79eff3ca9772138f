#!/bin/bash
# The CPU restatement (oracle/gsi_oracle.c) and the shipped pipeline.cpp / api.cpp over the CPU reference backend
# under AddressSanitizer + UBSan.  CPU build only: GPU sanitizers are not available on the pool.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
export GSI_CPUREF_LIB=$PWD/oracle/_build/asan/libgsi_cpuref.so GSI_ORACLE_C_LIB=$PWD/oracle/_build/asan/libgsi_oracle.so
# libstdc++ rides along so that ASan's __cxa_throw interceptor resolves inside the python process
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" \
  ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
  python -m pytest tests/test_cpuref_pipeline.py tests/test_oracle_c.py tests/test_cabi_symbols.py -x -q "$@"
