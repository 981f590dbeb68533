set -x
cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_b_pytest.txt
cat gpurun_out/r06_b_pytest.txt
for sc in ellipsoid template_like; do
  echo "== $sc" >> gpurun_out/r06_b_sweep.txt
  timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "0 2 3 4 6" --scene $sc --no-template-like >> gpurun_out/r06_b_sweep.txt 2>&1
done
cat gpurun_out/r06_b_sweep.txt
