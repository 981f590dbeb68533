#!/bin/bash
# bash tools/quick_profile.sh [bench args]: one bench line (value, min, p10) and the per-kernel averages of a short profiled run.  GPU box only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
timeout 120 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --frame-streams 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VALUE', d['value'], d['ms_per_step'], 'min', d['rounds']['min'], 'p10', d['rounds']['p10'])"
rm -rf /tmp/qprof
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qprof -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-profile --frame-streams 0 --min-seconds 0 "$@" > /tmp/qprof.log 2>&1 || tail -5 /tmp/qprof.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/qprof/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
steps=max(int(r['Calls']) for r in rows if 'k_render' in r['Name'])
tot=0
for r in rows:
    per=float(r['TotalDurationNs'])/1e3/steps; tot+=per
    if per>=2.0: print(f"{r['Name'].split('(')[0][:50]:50s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} per-step {per:7.1f}")
print('sum per step', round(tot,1))
PY
