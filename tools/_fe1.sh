cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_embed.py -x -q 2>&1 | tail -5
