#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace --stats run:  python tools/kstats.py <dir> <steps> [rows]"""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2])
rows_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    tot += float(r['TotalDurationNs'])
for r in rows[:rows_n]:
    n = r['Name'].split('(')[0][-58:]
    print(f"{n:60s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/steps/1000:8.1f} us/step  avg {float(r['AverageNs'])/1000:7.1f}")
print("sum of all kernels per step: %.1f us" % (tot / steps / 1000))
