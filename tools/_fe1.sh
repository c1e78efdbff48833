cd /root/repo
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['ms_per_step'], k['lstm_step_bwd'], k['lstm_dw'])"; }
echo dense; run
echo ragged; run --variant ragged
