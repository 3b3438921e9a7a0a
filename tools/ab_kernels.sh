#!/bin/bash
# Per-kernel averages (rocprofv3 kernel stats) of library variants built by tools/build_variant.sh, one process each. Usage: tools/ab_kernels.sh <tag> <regex> <variant>...
TAG=$1; RE=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in "$@"; do
  LIB=$ROOT/g4s_amd/lib_var/$v/libg4s_hip.so; [ "$v" = base ] && LIB=$ROOT/g4s_amd/lib/libg4s_hip.so
  export G4S_LIB=$LIB
  echo "== $v"; bash tools/prof_any.sh ${TAG}_$v tools/bench_spgemm.py --ef 3 --runs 3 2>&1 | grep -E "$RE" | cut -c1-40,72-140
done
