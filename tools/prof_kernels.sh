#!/bin/bash
# Per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats). Usage: tools/prof_kernels.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('pb_', 'spmv_', 'spgemm', 'hub_', 'elem_', 'dense_rows')):
        short = [t for t in ('pb_producer','pb_consumer','pb_scale','spmv_csr_adaptive','spmv_long_fixup') if t in n]
        short = short[0] if short else n[:40]
        print(f"{short:40s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f} us  min={float(r['MinNs'])/1e3:9.1f}  max={float(r['MaxNs'])/1e3:9.1f}")
PY
tail -1 $OUT/bench.log | cut -c1-400
