#!/bin/bash
# SpGEMM measurement run (one gpurun call): the bench line with the reference's call sequence on oneMKL beside it, rocprofv3 kernel stats,
# the per-dispatch timeline of one call, and the PMC passes. Results land in gpurun_out/<tag>/. Usage: tools/r02_spgemm_profile.sh <tag>
TAG=${1:-r02sp}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
echo "== spgemm bench"; timeout -k 10 900 python3 tools/bench_spgemm.py --ef 3 --runs 10 --mkl 14 > $O/spgemm_ef3.json 2> $O/spgemm_ef3.err; tail -c 2500 $O/spgemm_ef3.json
echo "== spgemm kernel stats"; timeout -k 10 400 bash tools/prof_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_kernel_stats.txt 2>&1; head -14 $O/spgemm_kernel_stats.txt
echo "== dispatches"; timeout -k 10 400 bash tools/prof_dispatches.sh $TAG "spgemm|row_flop|colmap|window_splits|classify|scatter|scan_|presort|row_size" tools/bench_spgemm.py --ef 3 --runs 1 > $O/spgemm_dispatches.txt 2>&1; tail -5 $O/spgemm_dispatches.txt
echo "== spgemm pmc"; timeout -k 10 900 bash tools/prof_pmc_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 2 > $O/spgemm_pmc.txt 2>&1; cp gpurun_out/pmc_$TAG/summary.json $O/spgemm_pmc_summary.json 2>/dev/null; tail -3 $O/spgemm_pmc.txt
