#!/bin/bash
# A/B builds of ONE source file: compiles <file> with extra flags and links it with the other objects of the regular build into
# g4s_amd/lib_var/<name>/libg4s_hip.so (load with G4S_LIB). Usage: tools/build_variant.sh <name> <file.hip> [flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; shift 2
O=$ROOT/g4s_amd/csrc/build_var/$NAME; L=$ROOT/g4s_amd/lib_var/$NAME
mkdir -p $O $L
make -s -C $ROOT/g4s_amd/csrc -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -I$ROOT/include -Wall -Wno-unused-result -c $ROOT/g4s_amd/csrc/$FILE -o $O/$FILE.o
# the variant says so in g4s_build_info (bench.py pairs stored PMC numbers with the regular build only)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -I$ROOT/include -Wall -Wno-unused-result \
    -DG4S_SPMV_KERNEL_HASH=\"$(python3 $ROOT/tools/kernel_hash.py)\" -DG4S_BUILD_VARIANT=\"$NAME\" -x hip -c $ROOT/g4s_amd/csrc/runtime.cpp -o $O/runtime.cpp.o
OBJS=$(ls $ROOT/g4s_amd/csrc/build/*.o | grep -v "/$FILE.o" | grep -v "/runtime.cpp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libg4s_hip.so $O/$FILE.o $O/runtime.cpp.o $OBJS
echo built $L/libg4s_hip.so
