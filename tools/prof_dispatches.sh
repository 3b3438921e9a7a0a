#!/bin/bash
# Every dispatch of the matching kernels in time order (rocprofv3 kernel trace). Usage: tools/prof_dispatches.sh <tag> <regex> <script> [args]
TAG=$1; shift
RE=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kd_$TAG
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1
python3 - "$OUT" "$RE" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rx = re.compile(sys.argv[2])
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name']
    if not rx.search(n): continue
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', n)
    short = (m.group(0) if m else n)[:60]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:10.3f} ms  {short:60s} grid={r['Grid_Size_X']:>9s} wg={r['Workgroup_Size_X']:>5s} lds={r['LDS_Block_Size']:>7s} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:10.1f} us")
PY
