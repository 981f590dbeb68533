cd /root/repo
: > gpurun_out/r06_h_sweep.txt
for sc in ellipsoid template_like; do
  for lib in "" build/exp/libgsr_prio1.so build/exp/libgsr_prio2.so; do
    echo "== $sc lib=$lib" >> gpurun_out/r06_h_sweep.txt
    GSR_LIB=${lib:+$PWD/$lib} timeout 900 bash tools/ab_env.sh GSR_CONT_CHUNKS "0" --scene $sc --no-template-like >> gpurun_out/r06_h_sweep.txt 2>&1
  done
done
awk '{print $1, $2, $3, $4, $5, $6, $7, $8, $9, $10}' gpurun_out/r06_h_sweep.txt
