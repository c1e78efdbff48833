cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['ms_per_step'], k['lstm_step_bwd'], k['lstm_dw'])"; }
echo "dense dw 4 stages"; run
echo "dense dw 3 stages"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_bf16_abl3.so run
echo "dense dw 5 stages"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_bf16_abl5.so run
echo "dense dw 4 stages"; run
echo "dense dw 5 stages"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_bf16_abl5.so run
FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_bf16_abl5.so timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
