#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the headline SpMV line, alternating, one process each, in one gpurun call. Usage: tools/ab_headline.sh <variant>...   ("base" = g4s_amd/lib)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for rep in 1 2 3; do
for v in "$@"; do
  LIB=$ROOT/g4s_amd/lib_var/$v/libg4s_hip.so; [ "$v" = base ] && LIB=$ROOT/g4s_amd/lib/libg4s_hip.so
  echo -n "$v rep $rep: "; G4S_LIB=$LIB python3 bench.py --no-also --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],'GEdges/s',d['ms_per_step'],'ms frac',d['roofline']['frac'],'kernel_ms',d['roofline']['kernel_ms'])"
done; done
