#!/bin/bash
# the CPU restatement (oracle/lsm_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the
# CPU tests that exercise nothing but the oracle.  No OpenMP in this build (libgomp's thread stacks confuse the leak checker and
# the tests run single-threaded anyway).  Usage: tools/oracle_asan.sh   (exit code = pytest's)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
ASAN_RT=$(gcc -print-file-name=libasan.so)
export LSM_ORACLE_LIB="$PWD/oracle/liblsm_oracle_asan.so"
# python itself is not instrumented: preload the runtime, do not fail on the interpreter's own leaks
export LD_PRELOAD="$ASAN_RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
exec python -m pytest -q -x -p no:cacheprovider tests/test_oracle_reference_tests.py tests/test_oracle_crosscheck.py tests/test_golden.py \
    tests/test_config1.py -m "not gpu" "$@"
