#!/bin/bash
# The two SpGEMM call forms side by side on configs[2] (one gpurun call): g4s_spgemm_symbolic + g4s_spgemm_numeric with the state the symbolic call leaves for the
# numeric call (default) and without it (G4S_SPGEMM_NO_CARRY=1), then the one-call form. Usage: tools/two_call_forms.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
two() { python3 tools/bench_spgemm.py --ef 3 --runs 5 --two-phase 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print('$1',d['symbolic_ms'],'+',d['numeric_ms'],'ms =',d['value'],'GFLOPS')"; }
two "two calls, carried   :"
G4S_SPGEMM_NO_CARRY=1 two "two calls, no carry  :"
python3 tools/bench_spgemm.py --ef 3 --runs 10 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print('one call             :',d['call_ms'],'ms =',d['value'],'GFLOPS')"
