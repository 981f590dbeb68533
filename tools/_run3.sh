set -x
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_c_pytest.txt
cat gpurun_out/r06_c_pytest.txt
: > gpurun_out/r06_c_sweep.txt
for sc in ellipsoid template_like; do
  echo "== $sc mode 1" >> gpurun_out/r06_c_sweep.txt
  GSR_CONT_MODE=1 timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "0 2 3 4" --scene $sc --no-template-like >> gpurun_out/r06_c_sweep.txt 2>&1
  echo "== $sc mode 2" >> gpurun_out/r06_c_sweep.txt
  GSR_CONT_MODE=2 timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "3" --scene $sc --no-template-like >> gpurun_out/r06_c_sweep.txt 2>&1
  echo "== $sc mode 1 grid" >> gpurun_out/r06_c_sweep.txt
  GSR_CONT_MODE=1 GSR_CONT_CHUNKS=3 timeout 900 bash tools/ab_env.sh GSR_CONT_GRID "256 512 1024" --scene $sc --no-template-like >> gpurun_out/r06_c_sweep.txt 2>&1
done
cut -c1-200 gpurun_out/r06_c_sweep.txt
