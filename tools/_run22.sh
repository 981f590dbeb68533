#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fast_blend_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_hitest.so bash tools/kstat_env.sh old k_render
bash tools/kstat_env.sh new k_render
done
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_hitest.so bash tools/kstat_env.sh old_t k_render --scene template_like
bash tools/kstat_env.sh new_t k_render --scene template_like
