#!/bin/bash
# rocprofv3 kernel averages of the gls:: kernels under the train workload:  bash tools/gls_kernel_times.sh
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/glsk
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/glsk -- python bench.py --workload train --no-cpu-baseline --no-kernel-profile --frame-streams 0 --steps 60 --warmup 15 --rounds 1 --min-seconds 0 > /tmp/glsk.log 2>&1
python - $(ls /tmp/glsk/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gls::" in r["Name"][:14] and int(r["Calls"]) >= 60:
        print("%-28s %7.2f us" % (r["Name"].split("(")[0].replace("void ", ""), float(r["AverageNs"]) / 1e3))
PY
