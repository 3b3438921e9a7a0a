#!/bin/bash
# Rebuild with variants of the streaming SpMV's compile-time knobs on the GPU box and time them back to back (same device).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in "" "-DG4S_STREAM_XCD_REMAP=0" ""; do
  touch g4s_amd/csrc/spmv.hip
  make -C g4s_amd/csrc -j4 EXTRA="$v" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  echo "variant [$v]"; python tools/ab_spmv.py --workloads lap5,banded,lap7 --variants 0 --rounds 5 --iters 40 2>/dev/null | grep flags | cut -c1-100
done
touch g4s_amd/csrc/spmv.hip; make -C g4s_amd/csrc -j4 > /dev/null 2>&1
