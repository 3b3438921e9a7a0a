#!/bin/bash
# Shape sweeps of the SpGEMM row classes through the G4S_SPGEMM_T_* switches (one gpurun call). Usage: tools/sweep_shapes.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { echo -n "$* : "; env "$@" python3 tools/bench_spgemm.py --ef 3 --runs 6 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print(d['call_ms'],'ms',d['value'],'GFLOPS')"; }
run X=1
run G4S_SPGEMM_T_NUM_MED=512 G4S_SPGEMM_T_NUM_LARGE=512 G4S_SPGEMM_T_NUM_M2=512
run G4S_SPGEMM_T_NUM_MED=1024 G4S_SPGEMM_T_NUM_LARGE=1024 G4S_SPGEMM_T_NUM_M2=1024
run G4S_SPGEMM_T_SYM_MED=512
run G4S_SPGEMM_T_SYM_MED=1024
run G4S_SPGEMM_T_SYM_LARGE=512 G4S_SPGEMM_T_SYM_WINDOW=512
run G4S_SPGEMM_T_NUM_M3=512 G4S_SPGEMM_M3_CUT=0
run G4S_SPGEMM_T_NUM_M3=512 G4S_SPGEMM_M3_CUT=16384
run G4S_SPGEMM_T_NUM_M3=256 G4S_SPGEMM_M3_CUT=8192
run X=1
