#!/bin/bash
# A/B runs of the tile-blocked SpMV (one bench.py call per setting, same box). Usage: tools/sweep_tb.sh "VAR=val VAR=val" "VAR=val" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
for setting in "$@"; do
  out=$(env $setting timeout -k 10 120 python3 $ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1)
  ms=$(echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])" 2>/dev/null || echo "FAILED: $out")
  echo "[$setting] kernel_ms frac = $ms" | tee -a $ROOT/gpurun_out/sweep_tb.log
done
