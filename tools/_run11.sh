cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 bash tools/ab_env.sh GLS_L1_FUSED "0 1" --no-template-like 2>&1 | cut -c1-420
