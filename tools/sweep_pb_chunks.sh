#!/bin/bash
# Work-item sizes of the blocked SpMV (G4S_PB_PCHUNK entries per producer item, G4S_PB_CCHUNK micro-runs per consumer item) on the headline line, one process each,
# the default first and last, in one gpurun call. Usage: tools/sweep_pb_chunks.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { echo -n "$* : "; env "$@" python3 bench.py --no-also --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'],'ms',d['value'],'GEdges/s')"; }
run X=1
run G4S_PB_PCHUNK=32768
run G4S_PB_PCHUNK=65536
run G4S_PB_PCHUNK=98304
run G4S_PB_PCHUNK=196608
run G4S_PB_PCHUNK=262144
run X=1
run G4S_PB_CCHUNK=32768
run G4S_PB_CCHUNK=65536
run G4S_PB_CCHUNK=262144
run G4S_PB_PCHUNK=65536 G4S_PB_CCHUNK=65536
run X=1
