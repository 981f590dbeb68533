cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
: > gpurun_out/r06_f_sweep.txt
for sc in ellipsoid template_like; do
  echo "== $sc" >> gpurun_out/r06_f_sweep.txt
  timeout 900 bash tools/ab_env.sh GSR_RANK_ILV "-1 0 8" --scene $sc --no-template-like >> gpurun_out/r06_f_sweep.txt 2>&1
done
awk '{print $1, $2, $3, $4, $5, $6, $7, $8, $9, $10}' gpurun_out/r06_f_sweep.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_f_bench_cfg3.json 2>gpurun_out/r06_f_bench_cfg3.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_f_bench_cfg3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['template_like'])
PY
