#!/bin/bash
# The CPU test suite (-m "not gpu": oracle vs golden vectors, host logic, C-ABI symbols, gloo world-2/3) on the sanitizer
# builds of the oracle and of the C host layer (AddressSanitizer + UBSan; CPU only -- GPU ASan is not available here).
#   bash tools/run_cpu_sanitized.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
make -s -j8 asan
export DFL_ORACLE_LIB=$PWD/oracle/liboracle_asan.so
export DFL_LIB=$PWD/dedflow_amd/libdedflow_asan.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
