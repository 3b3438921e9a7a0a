#!/bin/bash
# Round-4 measurement run (one gpurun call): the bench line (headline + also + also_banded + also_spgemm), rocprofv3 kernel stats and PMC passes of the headline
# command (--no-also: the SpMV launches only) with the kernel-source hash beside the traffic number, rocprofv3 kernel stats + PMC of the SpGEMM call.
# Results land in gpurun_out/<tag>/; copy what is to be judged into profiles/. Usage: tools/r04_profile.sh [tag]
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
echo "== bench"; timeout -k 10 500 python3 bench.py > $O/bench_rmat.json 2> $O/bench_rmat.err; tail -c 700 $O/bench_rmat.json; echo
echo "== kernel stats"; timeout -k 10 300 bash tools/prof_kernels.sh $TAG --no-also > $O/kernel_stats.txt 2>&1; grep -E "pb_|spmv" $O/kernel_stats.txt
cp gpurun_out/kt_$TAG/*/*kernel_stats.csv $O/bench_rmat_kernel_stats.csv 2>/dev/null
echo "== pmc"; timeout -k 10 600 bash tools/prof_pmc.sh $TAG --no-also > $O/pmc.txt 2>&1; cp gpurun_out/pmc_$TAG/summary.json $O/spmv_rmat_pmc_summary.json 2>/dev/null; tail -3 $O/pmc.txt
python3 - "$O" <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import kernel_hash
d = json.load(open(os.path.join(sys.argv[1], "spmv_rmat_pmc_summary.json")))
tot = 0
for k in ("pb_prepare", "pb_producer", "pb_consumer"):
    tot += d[k]["fetch_bytes_x2"] + d[k]["write_bytes"]
out = {"workload": "rmat", "n_gpus": 1, "spmv_path": 1, "hbm_bytes_per_launch": int(tot), "source": os.path.basename(sys.argv[1]) + "_spmv_rmat_pmc_summary.json",
       "formula": "sum over pb_prepare, pb_producer, pb_consumer of FETCH_SIZE x 2 + WRITE_SIZE (KiB -> bytes)", "kernel_sources_sha256": kernel_hash.spmv_kernel_hash()}
json.dump(out, open(os.path.join(sys.argv[1], "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out))
PY
echo "== spgemm kernel stats"; timeout -k 10 400 bash tools/prof_any.sh ${TAG}sp tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_kernel_stats.txt 2>&1; head -20 $O/spgemm_kernel_stats.txt
echo "== spgemm pmc"; timeout -k 10 600 bash tools/prof_pmc_any.sh ${TAG}sp tools/bench_spgemm.py --ef 3 --runs 2 > $O/spgemm_pmc.txt 2>&1; cp gpurun_out/pmc_${TAG}sp/summary.json $O/spgemm_pmc_summary.json 2>/dev/null; tail -2 $O/spgemm_pmc.txt
