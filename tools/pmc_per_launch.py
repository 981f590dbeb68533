#!/usr/bin/env python3
"""rocprofv3 --pmc output (one directory per counter pass) -> per-kernel per-launch averages, the JSON bench.py reads.

    python tools/pmc_per_launch.py out.json  <dir of the FETCH_SIZE pass>  <dir of the WRITE_SIZE pass>

Each pass is its own run of  rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d <dir> -- python bench.py ...
(never combined with sys/runtime/hip tracing).  Values are the counters' native unit (KB) summed over the rows rocprofv3
emits for one dispatch and averaged over the dispatches of a kernel."""
import collections
import csv
import glob
import json
import sys

out_path, dirs = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(dict)
for d in dirs:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    per_dispatch = collections.defaultdict(float)   # (kernel, counter, dispatch id) -> value
    for f in files:
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: [0.0, 0])
    for (k, c, _), v in per_dispatch.items():
        a = agg[(k, c)]
        a[0] += v
        a[1] += 1
    for (k, c), (tot, n) in agg.items():
        short = k.split("(")[0]
        if not any(t in short for t in ("gsr::", "gab::", "gls::")):
            continue
        res[short][f"{c}_KB_per_launch"] = tot / n
        res[short]["launches"] = n
json.dump(dict(sorted(res.items())), open(out_path, "w"), indent=1)
print(out_path, len(res), "kernels")
