#!/bin/bash
# SpGEMM iteration loop (one gpurun call): parity tests, bench line, section profiles (needs g4s_amd/lib_prof built with EXTRA=-DG4S_PROFILE_BIG), kernel stats.
# Usage: tools/r04_spgemm_quick.sh <tag> [notest]
TAG=${1:-r04q}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
if [ "$2" != "notest" ]; then
  timeout -k 10 700 python3 -m pytest tests/test_spgemm_gpu.py tests/test_mkl_pin_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
  tail -2 $O/pytest.txt
fi
python3 tools/bench_spgemm.py --ef 3 --runs 10 > $O/bench.json 2> $O/bench.err; python3 -c "import json;d=json.load(open('$O/bench.json'));print('one call',d['call_ms'],'ms',d['value'],'GFLOPS')"
python3 tools/bench_spgemm.py --ef 3 --runs 5 --two-phase > $O/bench2.json 2>> $O/bench.err; python3 -c "import json;d=json.load(open('$O/bench2.json'));print('two calls',d['symbolic_ms'],'+',d['numeric_ms'],'ms',d['value'],'GFLOPS')"
if [ -f g4s_amd/lib_prof/libg4s_hip.so ]; then
  G4S_LIB=g4s_amd/lib_prof/libg4s_hip.so python3 tools/sym_prof.py > $O/sym_prof.txt 2>&1; grep -v amdgpu.ids $O/sym_prof.txt
  G4S_LIB=g4s_amd/lib_prof/libg4s_hip.so python3 tools/big_prof.py > $O/big_prof.txt 2>&1; grep -v amdgpu.ids $O/big_prof.txt
fi
bash tools/prof_any.sh $TAG tools/bench_spgemm.py --ef 3 --runs 3 > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
