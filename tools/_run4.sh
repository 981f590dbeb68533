set -x
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_d_pytest.txt
cat gpurun_out/r06_d_pytest.txt
: > gpurun_out/r06_d_sweep.txt
for sc in ellipsoid template_like; do
  echo "== $sc off" >> gpurun_out/r06_d_sweep.txt
  timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "0" --scene $sc --no-template-like >> gpurun_out/r06_d_sweep.txt 2>&1
  for lib in "" build/exp/libgsr_w4.so build/exp/libgsr_w16.so; do
    echo "== $sc mode 2 lib=$lib" >> gpurun_out/r06_d_sweep.txt
    GSR_LIB=${lib:+$PWD/$lib} GSR_CONT_MODE=2 timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "2 3 4" --scene $sc --no-template-like >> gpurun_out/r06_d_sweep.txt 2>&1
  done
done
awk '{print $1, $2, $3, $4, $5, $6, $7, $8, $9, $10}' gpurun_out/r06_d_sweep.txt
