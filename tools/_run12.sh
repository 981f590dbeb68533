cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 bash tools/ab_env.sh GLS_L1_BLOCKS "64 128 255" --no-template-like 2>&1 | awk '{print $1,$2,$(NF-1),$NF}'
