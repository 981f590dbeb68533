#!/usr/bin/env python3
"""rocprofv3 --pmc output (one directory per counter pass) -> per-kernel per-launch averages, the JSON bench.py reads.

    python tools/pmc_per_launch.py out.json  <dir of the FETCH_SIZE pass>  <dir of the WRITE_SIZE pass>

Each pass is its own run of  rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d <dir> -- python bench.py ...
(never combined with sys/runtime/hip tracing).  Values are the counters' native unit (KB) summed over the rows rocprofv3
emits for one dispatch and averaged over the dispatches of a kernel."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

out_path, dirs = sys.argv[1], sys.argv[2:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def provenance():
    """What these numbers were measured on: the commit (when the tree carries a .git; gpurun snapshots do not -- GRAFT_COMMIT is the
    fallback), a digest of each native library as built, and the box (GPU name, the host's logical CPU count).  bench.py compares the
    library digests with the libraries it runs and marks `traffic` stale on a mismatch."""
    meta = {"commit": os.environ.get("GRAFT_COMMIT", "")}
    try:
        meta["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip() or meta["commit"]
    except Exception:   # noqa: BLE001
        pass
    libs = {}
    for name in ("libgsr_hip.so", "libgab_hip.so", "libgls_hip.so"):
        path = os.path.join(ROOT, "gaussianavatars_amd", name)
        if os.path.exists(path):
            libs[name] = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    meta["libraries"] = libs
    try:
        import torch

        meta["gpu"] = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
    except Exception:   # noqa: BLE001
        meta["gpu"] = None
    meta["host_cpus"] = os.cpu_count()
    return meta


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
out = dict(sorted(res.items()))
out["_meta"] = provenance()
json.dump(out, open(out_path, "w"), indent=1)
print(out_path, len(res), "kernels")
