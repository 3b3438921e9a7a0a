#!/bin/bash
# Round-5 measurement run, in two gpurun calls (each well inside the time limit):
#   tools/r05_profile.sh spmv  [tag]   the bench line, rocprofv3 kernel stats and PMC passes of the headline command (SpMV launches only), the traffic number with the
#                                      hash of the kernel sources the LOADED library was built from, plan-build kernel stats and HIP API gaps of one create
#   tools/r05_profile.sh spgemm [tag]  SpGEMM bench line, kernel stats, per-dispatch timeline of one call, PMC passes, the two call forms, the rank kernel's section
#                                      timers (needs g4s_amd/lib_var/prof: tools/build_variant.sh prof spgemm.hip -DG4S_PROFILE_BIG)
# Results land in gpurun_out/<tag>/; copy what is to be judged into profiles/.
WHAT=${1:-spmv}
TAG=${2:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
if [ "$WHAT" = spmv ]; then
  echo "== bench"; timeout -k 10 500 python3 bench.py > $O/bench_rmat.json 2> $O/bench_rmat.err; tail -c 700 $O/bench_rmat.json; echo
  echo "== kernel stats"; timeout -k 10 300 bash tools/prof_kernels.sh $TAG --no-also > $O/kernel_stats.txt 2>&1; grep -E "pb_|spmv" $O/kernel_stats.txt
  cp gpurun_out/kt_$TAG/*/*kernel_stats.csv $O/bench_rmat_kernel_stats.csv 2>/dev/null
  echo "== pmc"; timeout -k 10 600 bash tools/prof_pmc.sh $TAG --no-also > $O/pmc.txt 2>&1; cp gpurun_out/pmc_$TAG/summary.json $O/spmv_rmat_pmc_summary.json 2>/dev/null; tail -3 $O/pmc.txt
  python3 - "$O" <<'PY'
import json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from g4s_amd import capi
binfo = dict(kv.split("=", 1) for kv in capi.load().g4s_build_info().decode().split(";") if "=" in kv)
d = json.load(open(os.path.join(sys.argv[1], "spmv_rmat_pmc_summary.json")))
tot = sum(d[k]["fetch_bytes_x2"] + d[k]["write_bytes"] for k in ("pb_prepare", "pb_producer", "pb_consumer"))
out = {"workload": "rmat", "n_gpus": 1, "spmv_path": 1, "hbm_bytes_per_launch": int(tot), "source": os.path.basename(sys.argv[1]) + "_spmv_rmat_pmc_summary.json",
       "formula": "sum over pb_prepare, pb_producer, pb_consumer of FETCH_SIZE x 2 + WRITE_SIZE (KiB -> bytes)",
       "kernel_sources_sha256": binfo["spmv_kernel_sources_sha256"] + ("" if not binfo.get("variant") else "+variant:" + binfo["variant"])}
json.dump(out, open(os.path.join(sys.argv[1], "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out))
PY
  echo "== plan build"; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/plan_trace -- python3 $ROOT/tools/plan_create_trace.py > $O/plan_create.txt 2>&1; cd $ROOT
  grep create $O/plan_create.txt; python3 tools/plan_trace_report.py $O/plan_trace > $O/plan_api_gaps.txt 2>&1; cat $O/plan_api_gaps.txt
  cp $O/plan_trace/*/*kernel_stats.csv $O/plan_create_kernel_stats.csv 2>/dev/null
else
  echo "== spgemm bench"; timeout -k 10 300 python3 tools/bench_spgemm.py --ef 3 --runs 10 > $O/spgemm_ef3.json 2> $O/spgemm_ef3.err; tail -c 1500 $O/spgemm_ef3.json; echo
  echo "== spgemm kernel stats"; timeout -k 10 300 bash tools/prof_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 3 > $O/spgemm_kernel_stats.txt 2>&1; head -26 $O/spgemm_kernel_stats.txt
  echo "== dispatches"; timeout -k 10 300 bash tools/prof_dispatches.sh $TAG "." tools/bench_spgemm.py --ef 3 --runs 1 > $O/spgemm_dispatches.txt 2>&1; tail -5 $O/spgemm_dispatches.txt
  echo "== two call forms"; timeout -k 10 300 bash tools/two_call_forms.sh > $O/spgemm_two_call.txt 2>&1; cat $O/spgemm_two_call.txt
  if [ -f g4s_amd/lib_var/prof/libg4s_hip.so ]; then echo "== rank kernel sections"; G4S_LIB=$ROOT/g4s_amd/lib_var/prof/libg4s_hip.so timeout -k 10 200 python3 tools/rank_prof.py 2>&1 | grep -v amdgpu.ids > $O/spgemm_rank_sections.txt; cat $O/spgemm_rank_sections.txt; fi
  echo "== spgemm pmc"; timeout -k 10 600 bash tools/prof_pmc_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 2 > $O/spgemm_pmc.txt 2>&1; cp gpurun_out/pmc_$TAG/summary.json $O/spgemm_pmc_summary.json 2>/dev/null; tail -3 $O/spgemm_pmc.txt
fi
