#!/bin/bash
# The CPU test-suite with the oracle built under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan).
# Python itself is not instrumented: libasan is preloaded, leak checking is off (CPython's arenas), everything else aborts on the
# first finding.  Usage: bash tools/run_oracle_asan.sh [pytest args]   (CPU only: GPU sanitizers are not available on this pool)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
# the multi-process gloo test spawns interpreters that would each need the preload: it has no oracle-specific code paths of its own
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  ORACLE_LIB=$PWD/oracle/liboracle_asan.so OMP_NUM_THREADS=2 \
  python -m pytest tests -q -m "not gpu" -x --deselect tests/test_distributed_gloo.py "$@"
