#!/bin/bash
# Round-2 measurement run (one gpurun call): the bench line, rocprofv3 kernel stats and PMC passes of the same command, the SpGEMM bench with the
# reference's call sequence on oneMKL beside it, and PMC passes for the SpGEMM kernels. Results land in gpurun_out/r02/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r02
mkdir -p $O
cd $ROOT
echo "== bench"; timeout -k 10 300 python3 bench.py > $O/bench_rmat.json 2> $O/bench_rmat.err; tail -c 1500 $O/bench_rmat.json
echo "== kernel stats"; timeout -k 10 300 bash tools/prof_kernels.sh r02 > $O/kernel_stats.txt 2>&1; grep -E "pb_|spmv" $O/kernel_stats.txt
cp gpurun_out/kt_r02/*/*kernel_stats.csv $O/bench_rmat_kernel_stats.csv 2>/dev/null
echo "== pmc"; timeout -k 10 600 bash tools/prof_pmc.sh r02 > $O/pmc.txt 2>&1; cp gpurun_out/pmc_r02/summary.json $O/spmv_rmat_pmc_summary.json 2>/dev/null
echo "== spgemm"; timeout -k 10 900 python3 tools/bench_spgemm.py --ef 3 --runs 10 --mkl 14 > $O/spgemm_ef3.json 2> $O/spgemm_ef3.err; tail -c 2500 $O/spgemm_ef3.json
echo "== spgemm kernel stats"; timeout -k 10 400 bash tools/prof_any.sh r02sp tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_kernel_stats.txt 2>&1; head -14 $O/spgemm_kernel_stats.txt
echo "== spgemm pmc"; timeout -k 10 900 bash tools/prof_pmc_any.sh r02sp tools/bench_spgemm.py --ef 3 --runs 2 > $O/spgemm_pmc.txt 2>&1; cp gpurun_out/pmc_r02sp/summary.json $O/spgemm_pmc_summary.json 2>/dev/null; tail -3 $O/spgemm_pmc.txt
