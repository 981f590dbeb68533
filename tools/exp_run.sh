#!/bin/bash
# Per-kernel event timings of bench.py for a list of library variants:  bash tools/exp_run.sh "base noatomic ..." [bench args]
VARS=$1; shift
for v in $VARS; do
  if [ $v = base ]; then L=""; else L="GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_$v.so"; fi
  env $L python bench.py --steps 40 --warmup 10 --rounds 1 --min-seconds 0 --no-cpu-baseline --frame-streams 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['all_kernels']
print('$v', d['value'], ' '.join(f\"{n[2:]}={v['avg_us']}\" for n,v in k.items()))"
done
