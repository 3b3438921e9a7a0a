#!/bin/bash
# Round-4 SpGEMM measurement run (one gpurun call): bench line (no MKL leg), rocprofv3 kernel stats, per-dispatch timeline of one call, PMC passes.
# Results land in gpurun_out/<tag>/. Usage: tools/r04_spgemm_profile.sh <tag>
TAG=${1:-r04sp}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
echo "== spgemm bench"; timeout -k 10 300 python3 tools/bench_spgemm.py --ef 3 --runs 10 > $O/spgemm_ef3.json 2> $O/spgemm_ef3.err; tail -c 1500 $O/spgemm_ef3.json
echo "== spgemm kernel stats"; timeout -k 10 300 bash tools/prof_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_kernel_stats.txt 2>&1; head -24 $O/spgemm_kernel_stats.txt
echo "== dispatches"; timeout -k 10 300 bash tools/prof_dispatches.sh $TAG "spgemm|row_flop|colmap|window_splits|classify|scatter|scan_|presort|row_size|chunk_splits|radix|compact" tools/bench_spgemm.py --ef 3 --runs 1 > $O/spgemm_dispatches.txt 2>&1; tail -5 $O/spgemm_dispatches.txt
echo "== spgemm pmc"; timeout -k 10 600 bash tools/prof_pmc_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 2 > $O/spgemm_pmc.txt 2>&1; cp gpurun_out/pmc_$TAG/summary.json $O/spgemm_pmc_summary.json 2>/dev/null; tail -3 $O/spgemm_pmc.txt
