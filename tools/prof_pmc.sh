#!/bin/bash
# PMC passes for the SpMV bench (one counter group per run, as MI355X_MICROARCH.md §rocprofv3 PMC slots requires:
# FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2). Usage: tools/prof_pmc.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; }
done
python3 $ROOT/tools/pmc_summary.py $OUT
