cd /root/repo
for sc in ellipsoid template_like; do
  GSR_CONT_CHUNKS=0 bash tools/_kstat.sh ${sc}_off --scene $sc
  for c in 3 4; do
    GSR_CONT_MODE=1 GSR_CONT_CHUNKS=$c bash tools/_kstat.sh ${sc}_m1_c$c --scene $sc
    GSR_CONT_MODE=2 GSR_CONT_CHUNKS=$c bash tools/_kstat.sh ${sc}_m2w8_c$c --scene $sc
    GSR_LIB=$PWD/build/exp/libgsr_w4.so GSR_CONT_MODE=2 GSR_CONT_CHUNKS=$c bash tools/_kstat.sh ${sc}_m2w4_c$c --scene $sc
    GSR_LIB=$PWD/build/exp/libgsr_w16.so GSR_CONT_MODE=2 GSR_CONT_CHUNKS=$c bash tools/_kstat.sh ${sc}_m2w16_c$c --scene $sc
  done
done
