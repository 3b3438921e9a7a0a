#!/bin/bash
# CPU test suite against the sanitizer builds:
#   oracle/_asan/liboracle.so        gcc  -fsanitize=address,undefined   (make -C oracle asan)
#   g4s_amd/lib_asan/libg4s_hip.so   hipcc host code -fsanitize=address,undefined, device code untouched (make -C g4s_amd/csrc asan)
# python is not instrumented, so the sanitizer runtime is preloaded. Leak checking is off (the interpreter leaks by design).
# usage: tools/run_sanitized_cpu_tests.sh [oracle|lib|both] [pytest args...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WHAT=${1:-both}; shift || true
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ "$WHAT" = oracle ] || [ "$WHAT" = both ]; then
  make -C "$ROOT/oracle" asan
  echo "== oracle under ASan + UBSan"
  LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" G4S_ORACLE_SO="$ROOT/oracle/_asan/liboracle.so" \
    python -m pytest "$ROOT/tests/test_oracle_cpu.py" "$ROOT/tests/test_mkl_pin_cpu.py" "$ROOT/tests/test_dist_cpu.py" -x -q -m "not gpu" -p no:cacheprovider "$@"
fi
if [ "$WHAT" = lib ] || [ "$WHAT" = both ]; then
  make -C "$ROOT/g4s_amd/csrc" asan -j4
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  echo "== libg4s_hip.so host code under ASan + UBSan (runtime $RT)"
  LD_PRELOAD="$RT" G4S_LIB="$ROOT/g4s_amd/lib_asan/libg4s_hip.so" \
    python -m pytest "$ROOT/tests/test_dist_cpu.py" "$ROOT/tests/test_host_callbacks_cpu.py" "$ROOT/tests/test_capi_cpu.py" -x -q -m "not gpu" -p no:cacheprovider "$@"
fi
