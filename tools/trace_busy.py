"""GPU busy fraction and per-kernel launch gaps from a rocprofv3 kernel trace:  python tools/trace_busy.py <dir with *kernel_trace.csv> [skip_fraction]
Skips the first `skip_fraction` (default 0.4) of the dispatches (warm-up, allocator growth), then reports
sum(kernel durations) / (last end - first start) and the idle time that precedes each kernel name (mean, in us)."""
import csv, glob, sys, collections, json

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
rows = rows[int(len(rows) * skip):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gap = collections.defaultdict(lambda: [0, 0])
prev_end = rows[0][1]
for s, e, n in rows[1:]:
    g = gap[n]
    g[0] += max(0, s - prev_end)
    g[1] += 1
    prev_end = max(prev_end, e)
out = {"dispatches": len(rows), "busy_fraction": round(busy / span, 4), "span_ms": round(span / 1e6, 3),
       "idle_before_us": {n: round(g[0] / g[1] / 1e3, 2) for n, g in sorted(gap.items(), key=lambda kv: -kv[1][0])[:24]}}
print(json.dumps(out, indent=1))
