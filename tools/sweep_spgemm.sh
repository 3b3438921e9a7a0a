#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in "" "-DG4S_SPGEMM_BIG_LIMIT=65536" "-DG4S_SPGEMM_BIG_LIMIT=131072"; do
  touch g4s_amd/csrc/spgemm.hip
  make -C g4s_amd/csrc -j4 EXTRA="$v" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  echo "variant [$v]: $(python tools/bench_spgemm.py --ef 3 --runs 2 2>/dev/null | tail -1 | cut -c1-60) $(python tools/bench_spgemm.py --ef 3 --runs 2 2>/dev/null | tail -1 | grep -o '"symbolic_ms.*runs')"
done
touch g4s_amd/csrc/spgemm.hip; make -C g4s_amd/csrc -j4 > /dev/null 2>&1
