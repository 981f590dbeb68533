cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 bash tools/ab_env.sh GAA_FACE_SUM "rows kernel" --no-template-like 2>&1 | cut -c1-430
timeout 600 bash tools/ab_env.sh GAA_FACE_SUM "rows kernel" --workload cfg4 --no-template-like 2>&1 | cut -c1-430
