#!/bin/bash
# Per-kernel event timings of bench.py with an environment switch off / on:  bash tools/ab_env.sh GSR_RSCATTER_BALANCED "0 1" [bench args]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --steps 40 --warmup 10 --rounds 1 --min-seconds 0 --no-cpu-baseline --frame-streams 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['all_kernels']
print('$VAR=$v', d['value'], ' '.join(f\"{n[2:]}={v['avg_us']}\" for n,v in k.items()))"
done
