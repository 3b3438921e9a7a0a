#!/bin/bash
# Rebuild the library with each variant of the blocked-SpMV compile-time knobs ON THE GPU BOX and time them back to back
# (same device, same process layout): variants are comparable with each other, not with numbers from other calls.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in "" "-DG4S_PB_NT=0" "-DG4S_PB_PAIR_UNROLL=8" "-DG4S_PB_PAIR_UNROLL=2" "-DG4S_PB_CONS_UNROLL=4" "-DG4S_PB_CONS_UNROLL=1" ""; do
  touch g4s_amd/csrc/spmv_pb.hip
  make -C g4s_amd/csrc -j4 EXTRA="$v" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  echo "variant [$v]: $(python tools/ab_spmv.py --workloads rmat --variants 0 --rounds 5 --iters 40 2>/dev/null | tail -1 | cut -c1-110)"
done
touch g4s_amd/csrc/spmv_pb.hip; make -C g4s_amd/csrc -j4 > /dev/null 2>&1
