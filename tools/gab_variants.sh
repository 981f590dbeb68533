#!/bin/bash
# rocprofv3 kernel averages of the gab:: kernels for experiment builds of libgab_hip.so (build/exp/libgab_NAME.so):  bash tools/gab_variants.sh "base vpb16 ..."
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in $1; do
  if [ $v = base ]; then L=""; else L="GAB_LIB=$GRAFT_REPO_ROOT/build/exp/libgab_$v.so"; fi
  rm -rf /tmp/gv_$v
  env $L timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gv_$v -- python bench.py --no-cpu-baseline --no-kernel-profile --frame-streams 0 --steps 60 --warmup 15 --rounds 1 --min-seconds 0 > /tmp/gv_$v.log 2>&1
  python - $v $(ls /tmp/gv_$v/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
out = []
for r in rows:
    if r["Name"].startswith("gab::") or "gab::" in r["Name"][:12]:
        if int(r["Calls"]) >= 60: out.append("%s=%.2f" % (r["Name"].split("(")[0].replace("void ", "").replace("gab::", "")[:22], float(r["AverageNs"]) / 1e3))
print(sys.argv[1], " ".join(sorted(out)))
PY
done
