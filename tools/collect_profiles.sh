#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on a GPU box:  bash tools/collect_profiles.sh <tag>   (e.g. r02_a)
# kernel stats and the two PMC counters are separate runs (PMC is never combined with other trace domains).
set -u
TAG=${1:-r06_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-profile --frame-streams 0 --no-template-like"
cd $GRAFT_REPO_ROOT
for W in cfg3 cfg4 cfg5; do
  if [ $W = cfg3 ]; then S="--steps 60 --warmup 15 --rounds 1 --min-seconds 0"; else S="--steps 30 --warmup 8 --rounds 1 --min-seconds 0"; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -- $B --workload $W $S > $OUT/${W}_stats.log 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${W}_fetch -- $B --workload $W $S > $OUT/${W}_fetch.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${W}_write -- $B --workload $W $S > $OUT/${W}_write.log 2>&1
  python tools/pmc_per_launch.py $OUT/${TAG}_${W}_pmc_fetch_write_per_launch.json $OUT/${W}_fetch $OUT/${W}_write
  cp $OUT/${TAG}_${W}_pmc_fetch_write_per_launch.json profiles/   # (this box's copy of the tree: the bench lines below read roofline.traffic from the passes of THESE libraries)
  cp $(ls $OUT/${W}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_${W}_kernel_stats.csv
  rm -rf $OUT/${W}_fetch $OUT/${W}_write   # raw per-dispatch tables are large; the summaries are what is kept
  python tools/trace_busy.py $OUT/${W}_stats > $OUT/${TAG}_${W}_busy.json
  rm -f $OUT/${W}_stats/*/*kernel_trace.csv
done
# the second scene (round 6): the same step on the template-like head
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3_template_stats -- $B --workload cfg3 --scene template_like --steps 60 --warmup 15 --rounds 1 --min-seconds 0 > $OUT/cfg3_template_stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/cfg3_template_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_cfg3_template_kernel_stats.csv
rm -f $OUT/cfg3_template_stats/*/*kernel_trace.csv
timeout 200 python bench.py --scene template_like --steps 150 --warmup 30 --no-cpu-baseline --frame-streams 0 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_template_like.json
timeout 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3.json
timeout 200 python bench.py --workload cfg2 --steps 300 --warmup 40 --no-cpu-baseline --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg2.json
timeout 200 python bench.py --workload cfg4 --steps 100 --warmup 20 --no-cpu-baseline --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg4.json
timeout 200 python bench.py --workload cfg5 --steps 40 --warmup 8 --no-cpu-baseline --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg5.json
timeout 200 python bench.py --workload cfg5 --steps 40 --warmup 8 --no-cpu-baseline --frame-streams 0 --no-template-like --no-spatial-sort 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg5_as_generated.json
# the plain default line (cpu_baseline, frame_streams leg), the train workload, and the recorded step pinned / un-pinned beside the eager loop un-pinned
timeout 400 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default.json
timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_style.json
timeout 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --frame-streams 0 --no-template-like --no-spatial-sort 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_as_generated.json
timeout 300 python bench.py --graph --streams 4 --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_4lanes_shared.json
timeout 200 python bench.py --workload train --steps 100 --warmup 20 --no-cpu-baseline --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train.json
G="--steps 200 --warmup 30 --no-cpu-baseline --no-kernel-profile --frame-streams 0 --no-template-like"
timeout 200 python bench.py --graph $G 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_graph.json
timeout 200 python bench.py --graph --no-pin $G 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_graph_unpinned.json
timeout 200 python bench.py --no-pin $G 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_eager_unpinned.json
# round 5: the compiled host side against the Python twins on this box (same libraries, same kernels), host time per step, the reference's own scripts
GAA_NATIVE_HOST=0 timeout 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_python_host.json
GAA_NATIVE_HOST=0 timeout 200 python bench.py --no-pin --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-profile --frame-streams 0 --no-template-like 2>/dev/null | tail -1 > $OUT/${TAG}_bench_cfg3_python_host_unpinned.json
timeout 200 python tools/host_profile.py 300 > $OUT/${TAG}_host_profile.txt 2>&1
timeout 200 python tools/host_profile.py 300 --small >> $OUT/${TAG}_host_profile.txt 2>&1
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5) > $OUT/${TAG}_gputests.log
if [ -d _ref_scratch/reference ]; then
  (GAA_REF_TAG=${TAG}_ref timeout 900 python tools/ref_on_gpu.py run) > $OUT/${TAG}_ref_run.log 2>&1
  for f in gpurun_out/${TAG}_ref_*; do cp $f $OUT/; done
fi
ls -la $OUT
