#!/bin/bash
# A/B runs of the SpGEMM bench (one tools/bench_spgemm.py call per setting, same box). Usage: tools/sweep_spgemm.sh "VAR=val VAR=val" "" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
for setting in "$@"; do
  out=$(env $setting timeout -k 10 200 python3 $ROOT/tools/bench_spgemm.py --ef ${EF:-3} --runs ${RUNS:-5} 2>&1 | tail -1)
  ms=$(echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['call_ms'], d['value'])" 2>/dev/null || echo "FAILED: $out")
  echo "[$setting] call_ms GFLOPS = $ms" | tee -a $ROOT/gpurun_out/sweep_spgemm.log
done
