#!/bin/bash
# A/B of library variants built by tools/build_variant.sh: the SpGEMM call time and the per-kernel averages (rocprofv3) of each, alternating, in one gpurun call.
# Usage: tools/ab_variants.sh <tag> <variant> [<variant> ...]      ("base" = g4s_amd/lib)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O; cd $ROOT
for rep in 1 2; do
for v in "$@"; do
  LIB=$ROOT/g4s_amd/lib_var/$v/libg4s_hip.so; [ "$v" = base ] && LIB=$ROOT/g4s_amd/lib/libg4s_hip.so
  G4S_LIB=$LIB python3 tools/bench_spgemm.py --ef 3 --runs 10 > $O/bench_$v.json 2>> $O/err.txt
  python3 -c "import json;d=json.load(open('$O/bench_$v.json'));print('$v rep $rep: one call',d['call_ms'],'ms',d['value'],'GFLOPS')"
done; done
for v in "$@"; do
  LIB=$ROOT/g4s_amd/lib_var/$v/libg4s_hip.so; [ "$v" = base ] && LIB=$ROOT/g4s_amd/lib/libg4s_hip.so
  export G4S_LIB=$LIB
  echo "== $v"; bash tools/prof_any.sh ${TAG}_$v tools/bench_spgemm.py --ef 3 --runs 3 2>&1 | grep -E "numeric_big|symbolic_window|numeric_lds" | cut -c1-40,72-140
done
