cd /root/repo
: > gpurun_out/r06_g_sweep.txt
for sc in ellipsoid template_like; do
  echo "== $sc" >> gpurun_out/r06_g_sweep.txt
  timeout 900 bash tools/ab_env.sh GSR_RANK_BUCKET_SPLATS "64 128 256 512" --scene $sc --no-template-like >> gpurun_out/r06_g_sweep.txt 2>&1
done
awk '{print $1, $2, $3, $4, $5, $6, $7, $8, $9, $10}' gpurun_out/r06_g_sweep.txt
