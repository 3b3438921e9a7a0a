#!/usr/bin/env python3
"""Per-call kernel time table from a rocprofv3 kernel_stats.csv (tools/prof_any.sh output). Usage: tools/kstats.py <kt dir> [calls per kernel = 4]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = 0.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    m = re.search(r'(\w+(_kernel|Aligned|Buffer))(<[^(]*>)?', r['Name'])
    short = (m.group(0) if m else r['Name'])[:58]
    per = float(r['TotalDurationNs']) / div / 1e6
    print(f"{short:58s} launches/call={int(r['Calls']) / div:6.1f} ms/call={per:7.3f}")
print('all kernels, ms per call:', round(sum(float(r['TotalDurationNs']) for r in rows) / div / 1e6, 3))
