#!/bin/bash
# SQ counters of one kernel:  [PMC_BENCH_ARGS="--workload cfg4"] bash tools/pmc_kernel.sh <kernel substring> "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
K=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for C in "$@"; do
  i=$((i+1)); D=gpurun_out/pmck_$i; rm -rf $D
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python bench.py ${PMC_BENCH_ARGS:-} --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --frame-streams 0 > $D.log 2>&1
  python - "$D" "$K" <<'PY'
import csv, glob, sys, collections
d, k = sys.argv[1], sys.argv[2]
per = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            per[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
for c, v in sorted(per.items()):
    print(f"{c:28s} {v / max(1, len(disp[c])):16.1f} per launch  ({len(disp[c])} launches)")
PY
  rm -rf $D   # the raw per-dispatch tables are large (gpurun_out/ is capped at 64 MiB); the printed summary is what is kept
done
