#!/bin/bash
# Round-4 evidence run (one gpurun call) for the tables of DESIGN.md that are not kernel stats / PMC of the two headline commands: the SpMV byte table, the row-class
# structure of configs[2]'s product, the host phases of one SpGEMM call, the section timers of the window kernels (needs tools/build_variant.sh prof spgemm.hip
# -DG4S_PROFILE_BIG beforehand), the dispatch timeline of one call, single ranks of the 8-way slab, and the bench line with the traffic file of the same sources.
# Results land in gpurun_out/<tag>/. Usage: tools/r04_evidence.sh [tag]
TAG=${1:-r04ev}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
echo "== bench"; timeout -k 10 500 python3 bench.py > $O/bench_rmat.json 2> $O/bench_rmat.err; tail -c 600 $O/bench_rmat.json; echo
echo "== byte table"; timeout -k 10 300 python3 tools/byte_table.py > $O/spmv_byte_table.txt 2>&1; tail -12 $O/spmv_byte_table.txt
echo "== row classes"; timeout -k 10 300 python3 tools/row_hist.py --ef 3 > $O/spgemm_row_classes.txt 2>&1; tail -5 $O/spgemm_row_classes.txt
echo "== host phases"; G4S_DEBUG=1 timeout -k 10 300 python3 tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_host_phases.txt 2>&1; tail -30 $O/spgemm_host_phases.txt
if [ -f g4s_amd/lib_var/prof/libg4s_hip.so ]; then
  echo "== section timers"; (G4S_LIB=$ROOT/g4s_amd/lib_var/prof/libg4s_hip.so timeout -k 10 300 python3 tools/big_prof.py --ef 3; G4S_LIB=$ROOT/g4s_amd/lib_var/prof/libg4s_hip.so timeout -k 10 300 python3 tools/sym_prof.py --ef 3) > $O/spgemm_section_timers.txt 2>&1; tail -30 $O/spgemm_section_timers.txt
fi
echo "== dispatches"; timeout -k 10 300 bash tools/prof_dispatches.sh $TAG "spgemm|row_flop|entry_flop|colmap|window_splits|classify|scatter|scan_|chunk_splits|unit_|sym_|diff_|check_|row_start|sort_" tools/bench_spgemm.py --ef 3 --runs 1 > $O/spgemm_dispatches.txt 2>&1; tail -3 $O/spgemm_dispatches.txt
for r in 0 4; do echo "== slab rank $r"; timeout -k 10 300 bash tools/prof_any.sh ${TAG}slab$r tools/slab_kernels.py $r 8 > $O/slab_rank${r}_kernel_stats.txt 2>&1; head -8 $O/slab_rank${r}_kernel_stats.txt; done
