#!/bin/bash
# Round-3 measurement run (one gpurun call): the bench line (headline + also + also_spgemm), rocprofv3 kernel stats and PMC passes of the headline
# command (--no-also: the SpMV launches only), rocprofv3 kernel stats of the SpGEMM call. Results land in gpurun_out/r03/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03
mkdir -p $O
cd $ROOT
echo "== bench"; timeout -k 10 400 python3 bench.py > $O/bench_rmat.json 2> $O/bench_rmat.err; tail -c 600 $O/bench_rmat.json; echo
echo "== kernel stats"; timeout -k 10 300 bash tools/prof_kernels.sh r03 --no-also > $O/kernel_stats.txt 2>&1; grep -E "pb_|spmv" $O/kernel_stats.txt
cp gpurun_out/kt_r03/*/*kernel_stats.csv $O/bench_rmat_kernel_stats.csv 2>/dev/null
echo "== pmc"; timeout -k 10 600 bash tools/prof_pmc.sh r03 --no-also > $O/pmc.txt 2>&1; cp gpurun_out/pmc_r03/summary.json $O/spmv_rmat_pmc_summary.json 2>/dev/null; tail -3 $O/pmc.txt
echo "== spgemm kernel stats"; timeout -k 10 400 bash tools/prof_any.sh r03sp tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_kernel_stats.txt 2>&1; head -16 $O/spgemm_kernel_stats.txt
