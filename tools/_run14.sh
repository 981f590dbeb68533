cd /root/repo
for i in 1 2; do
for lib in "" build/exp/libgsr_lb5.so build/exp/libgsr_r05.so; do
  echo "== lib=$lib"
  GSR_LIB=${lib:+$PWD/$lib} timeout 600 bash tools/ab_env.sh GSR_FAST_BLEND "1" --no-template-like 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}'
done
done
timeout 600 bash tools/ab_env.sh GSR_RANK_ILV "-1 8" --scene template_like --no-template-like 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}'
