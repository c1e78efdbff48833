cd /root/repo
export PYTHONPATH=/root/repo
run() { timeout 300 python bench.py --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['ms_per_step'], k['lstm_step_bwd'], k['lstm_dw'])"; }
echo "dense product"; run
for b in 0 45 70; do echo "split dirs offset $b"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_abl$b.so run; done
echo "dense product"; run
FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_abl45.so timeout 300 python -m pytest tests/test_gpu_bf16.py -x -q -k "bilstm" 2>&1 | tail -2
