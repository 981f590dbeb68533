set -x
cd /root/repo
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_a_bench_cfg3.json 2> gpurun_out/r06_a_bench_cfg3.err
tail -c 3000 gpurun_out/r06_a_bench_cfg3.json
for sc in ellipsoid template_like; do
  GAA_BENCH_SCENE=$sc GSR_LIB=$PWD/gaussianavatars_amd/libgsr_timeline.so python tools/bwd_timeline.py > gpurun_out/r06_a_timeline_$sc.txt 2>&1
  GAA_BENCH_SCENE=$sc python tools/stream_dump.py cfg3 && mv gpurun_out/stream_dump_cfg3.npz gpurun_out/stream_dump_cfg3_$sc.npz
done
grep -A40 FORWARD gpurun_out/r06_a_timeline_template_like.txt
