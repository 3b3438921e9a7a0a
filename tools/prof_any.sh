#!/bin/bash
# rocprofv3 kernel stats of any python tool. Usage: tools/prof_any.sh <tag> <script> [args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:22]:
    n = r['Name']
    i = n.find('::', n.find('namespace)')) if 'namespace)' in n else -1
    short = (n[i + 2:] if i >= 0 else n)[:70]
    print(f"{short:70s} calls={r['Calls']:>5s} total={float(r['TotalDurationNs'])/1e6:9.2f} ms avg={float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s}%")
PY
tail -2 $OUT/run.log | cut -c1-600
