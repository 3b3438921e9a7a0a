#!/bin/bash
# SpGEMM call time of library variants (tools/build_variant.sh), alternating, in one gpurun call. Usage: tools/ab_libs.sh <reps> <variant> [<variant> ...]   ("base" = g4s_amd/lib)
REPS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for rep in $(seq 1 $REPS); do
for v in "$@"; do
  LIB=$ROOT/g4s_amd/lib_var/$v/libg4s_hip.so; [ "$v" = base ] && LIB=$ROOT/g4s_amd/lib/libg4s_hip.so
  G4S_LIB=$LIB timeout -k 10 200 python3 tools/bench_spgemm.py --ef 3 --runs 6 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$v', 'rep $rep:', d['call_ms'], 'ms', d['value'], 'GFLOPS')"
done; done
