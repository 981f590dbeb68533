#!/bin/bash
# Builds experiment variants of libgsr_hip.so side by side:  bash tools/exp_build.sh NAME "-DGSR_EXP_FOO -DGSR_EXP_BAR=3"
# -> build/exp/libgsr_NAME.so   (run with GSR_LIB=build/exp/libgsr_NAME.so python bench.py ...)
set -e
NAME=$1; DEFS=$2
cd "$(dirname "$0")/../gaussianavatars_amd/csrc"
OUT=../../build/exp; mkdir -p $OUT
C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -munsafe-fp-atomics $DEFS"
/opt/rocm/bin/hipcc $C -ffp-contract=off ${FWD_FLAGS--fno-slp-vectorize} -c gsr_forward.hip -o $OUT/fwd_$NAME.o &
/opt/rocm/bin/hipcc $C -ffp-contract=fast -c gsr_backward.hip -o $OUT/bwd_$NAME.o &
/opt/rocm/bin/hipcc $C -ffp-contract=off -c gsr_api.hip -o $OUT/api_$NAME.o &
/opt/rocm/bin/hipcc $C -ffp-contract=off -c gsr_binning.hip -o $OUT/bin_$NAME.o &
/opt/rocm/bin/hipcc $C -ffp-contract=off -c gsr_rank.hip -o $OUT/rank_$NAME.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgsr_$NAME.so $OUT/fwd_$NAME.o $OUT/bwd_$NAME.o $OUT/api_$NAME.o $OUT/bin_$NAME.o $OUT/rank_$NAME.o
rm -f $OUT/*_$NAME.o
echo built $OUT/libgsr_$NAME.so
