cd /root/repo
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_gpu_embed.py tests/test_gpu_model.py tests/test_gpu_feed.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --front-end 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
timeout 300 python bench.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
