#!/bin/bash
# usage: _kstat.sh TAG [bench args]   (env selects the variant) -> prints the k_render rows of rocprofv3's kernel stats
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-profile --frame-streams 0 --no-template-like --steps 40 --warmup 10 --rounds 1 --min-seconds 0 "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
python - "$f" "$TAG" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'k_render' in r['Name'] and 'bwd' not in r['Name']:
        print(sys.argv[2], r['Name'].split('(')[0][-40:], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1000))
PY
rm -rf $OUT
