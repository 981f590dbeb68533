#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts:  bash tools/pmc_calibrate.sh <tag>   -> gpurun_out/<tag>_pmc_calibration.json
# (two PMC passes of the same binary, each with --kernel-trace only; see tools/pmc_calibrate.hip)
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics $GRAFT_REPO_ROOT/tools/pmc_calibrate.hip -o /tmp/pmc_calibrate || exit 1
/tmp/pmc_calibrate 2 > $OUT/${TAG}_cal_known.json
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/cal_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cal_$C -- /tmp/pmc_calibrate 2 > $OUT/cal_$C.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cal_stats -- /tmp/pmc_calibrate 2 > $OUT/cal_stats.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
known = json.load(open(f"{out}/{tag}_cal_known.json"))
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float)
    for f in glob.glob(f"{out}/cal_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])] += float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for (k, _), v in per.items():
        agg[k].append(v)
    for k, v in agg.items():
        res[k][c + "_bytes_per_launch"] = 1024.0 * sum(v) / len(v)
dur = {}
for f in glob.glob(f"{out}/cal_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0]] = float(r["AverageNs"]) / 1e3
table = {}
for k, req in known["requested"].items():
    m = res.get(k, {})
    row = dict(requested_bytes=req, fetch_bytes=m.get("FETCH_SIZE_bytes_per_launch"), write_bytes=m.get("WRITE_SIZE_bytes_per_launch"), avg_us=dur.get(k))
    if row["fetch_bytes"]: row["fetch_over_requested"] = row["fetch_bytes"] / req
    if row["write_bytes"]: row["write_over_requested"] = row["write_bytes"] / req
    if row["avg_us"]: row["requested_GBs"] = req / row["avg_us"] / 1e3
    for lk, lv in known["line_footprint"].items():
        if lk.startswith(k) and row["fetch_bytes"]:
            row["fetch_over_" + lk.split("_")[-1] + "_lines"] = row["fetch_bytes"] / lv
    table[k] = row
json.dump(dict(known=known, kernels=table), open(f"{out}/{tag}_pmc_calibration.json", "w"), indent=1)
print(json.dumps(table, indent=1))
PY
rm -rf $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE $OUT/cal_stats
