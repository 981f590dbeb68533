#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fast_blend_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_norow.so bash tools/kstat_env.sh norow k_render_bwd
bash tools/kstat_env.sh rowsum k_render_bwd
done
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_norow.so bash tools/kstat_env.sh norow_t k_render_bwd --scene template_like
bash tools/kstat_env.sh rowsum_t k_render_bwd --scene template_like
