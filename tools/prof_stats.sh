#!/bin/bash
# rocprofv3 kernel stats of a short bench run:  bash tools/prof_stats.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.csv + summary on stdout
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 40 --warmup 10 --rounds 1 --min-seconds 0 --no-cpu-baseline --no-kernel-profile "$@" > $OUT/stats.log 2>&1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
python tools/kstats.py $OUT/stats 50 30
rm -rf $OUT/stats
