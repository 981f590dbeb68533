cd /root/repo
timeout 900 bash tools/ab_env.sh GSR_RANK_ILV "0 8 64" --workload cfg5 2>&1 | cut -c1-300
timeout 900 bash tools/ab_env.sh GSR_RANK_ILV "0 8" --workload cfg4 --no-template-like 2>&1 | cut -c1-300
