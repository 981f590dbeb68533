#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py under the caller's environment, printing the rows whose kernel name matches PATTERN:
#   GSR_CONT_CHUNKS=3 bash tools/kstat_env.sh TAG k_render [bench args]        (on the GPU box; the A/B tables of profiles/r06_exp_*.txt)
TAG=$(echo "$1" | tr -c "A-Za-z0-9_\n" "_"); PATTERN=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-profile --frame-streams 0 --no-template-like --steps 40 --warmup 10 --rounds 1 --min-seconds 0 "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
python - "$f" "$TAG" "$PATTERN" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in r['Name']:
        print(sys.argv[2], r['Name'].split('(')[0][-48:], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1000))
PY
rm -rf $OUT
