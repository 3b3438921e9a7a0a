#!/bin/bash
# FETCH_SIZE and L2 hit counters of the diagonal kernel on the 431^3 operator with the plane-sliced (G4S_SPMV_DIA_WALK=l) and the contiguous (default) XCD walk (one box, one call).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in sliced contig; do
  if [ $mode = contig ]; then export G4S_SPMV_DIA_WALK=c; else export G4S_SPMV_DIA_WALK=l; fi
  O=$ROOT/gpurun_out/pmc2_$mode
  mkdir -p $O
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --pmc $grp --output-format csv -d $O/$name -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload lap7 --no-also > $O/$name.log 2>&1
  done
  python3 $ROOT/tools/pmc_summary.py $O > $O.txt 2>&1
  echo "$mode $(grep -h ms_per_step $O/FETCH_SIZE.log | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
