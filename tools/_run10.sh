cd /root/repo
timeout 600 python -m pytest tests/test_binding_gpu.py tests/test_model_pins.py tests/test_native_host_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -3
: > gpurun_out/r06_h_sweep.txt
for sc in ellipsoid template_like; do
  for lib in "" build/exp/libgsr_prio1.so build/exp/libgsr_prio2.so; do
    echo "== $sc lib=$lib" >> gpurun_out/r06_h_sweep.txt
    GSR_LIB=${lib:+$PWD/$lib} timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "0" --scene $sc --no-template-like >> gpurun_out/r06_h_sweep.txt 2>&1
  done
done
cat gpurun_out/r06_h_sweep.txt
