cd /root/repo
export PYTHONPATH=/root/repo
run() { timeout 200 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step']['lstm_step_bwd'])"; }
echo product; run
for b in 1; do echo pair $b; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_bf16_abl$b.so run; done
FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_bf16_abl1.so timeout 300 python -m pytest tests/test_gpu_bf16.py -x -q -k bilstm 2>&1 | tail -2
