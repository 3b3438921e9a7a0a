#!/bin/bash
# A/B of environment switches on the SpGEMM call, alternating, in one gpurun call. Usage: tools/ab_env.sh "VAR=1" ["VAR2=1" ...]   (X=1 = the default)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for rep in 1 2 3; do
for e in X=1 "$@"; do
  echo -n "$e rep $rep: "; env $e python3 tools/bench_spgemm.py --ef 3 --runs 10 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print(d['call_ms'],'ms',d['value'],'GFLOPS, min',d.get('min_ms'))"
done; done
