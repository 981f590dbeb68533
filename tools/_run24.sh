#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_nodirect.so bash tools/kstat_env.sh nodirect "k_render<"
bash tools/kstat_env.sh direct "k_render<"
done
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_nodirect.so bash tools/kstat_env.sh nodirect_t "k_render<" --scene template_like
bash tools/kstat_env.sh direct_t "k_render<" --scene template_like
GSR_LIB=$GRAFT_REPO_ROOT/build/exp/libgsr_nodirect.so bash tools/kstat_env.sh nodirect_5 "k_render<" --workload cfg5
bash tools/kstat_env.sh direct_5 "k_render<" --workload cfg5
