"""Prints the dispatch sequence of one frame from a rocprofv3 --kernel-trace directory: gap to the previous dispatch, duration,
queue, kernel.   python tools/trace_sequence.py <dir> [frames-from-the-end]"""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("gsr::k_preprocess(")]
i0 = idx[-back]
i1 = idx[-back + 1] if back > 1 else len(rows)
prev_end = None
for r in rows[max(0, i0 - 6):i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.2f %8.2f  q=%s %s" % (gap, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    prev_end = max(prev_end or 0, e)
