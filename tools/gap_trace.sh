#!/bin/bash
# bash tools/gap_trace.sh [bench args]: kernel-trace of a short bench run; prints, per frame, the summed kernel durations and the summed idle gaps
# between consecutive kernels (the GPU-side cost of launch boundaries).  Compare `--graph` with the eager loop.  GPU box only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gtrace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gtrace -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-profile --frame-streams 0 --min-seconds 0 --rounds 1 "$@" > /tmp/gtrace.log 2>&1 || tail -5 /tmp/gtrace.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/gtrace/**/*kernel_trace.csv', recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-28:]) for r in csv.DictReader(open(f))]
rows.sort()
# frames: from one k_flame_fused to the next, take the last 30
starts=[i for i,r in enumerate(rows) if 'k_flame_fused' in r[2]]
starts=starts[-31:]
dur=gap=0; n=0; pergap={}
for a,b in zip(starts[:-1],starts[1:]):
    fr=rows[a:b+1]
    for x,y in zip(fr[:-1],fr[1:]):
        dur+=x[1]-x[0]; g=y[0]-x[1]; gap+=g
        pergap[(x[2],y[2])]=pergap.get((x[2],y[2]),0)+g
    n+=1
print(f'frames {n}: kernel time {dur/n/1e3:.1f} us/frame, gaps {gap/n/1e3:.1f} us/frame, frame period {(rows[starts[-1]][0]-rows[starts[0]][0])/n/1e3:.1f} us')
for k,v in sorted(pergap.items(), key=lambda kv:-kv[1])[:24]:
    print(f'  {k[0]:>28s} -> {k[1]:<28s} {v/n/1e3:6.2f} us')
PY
