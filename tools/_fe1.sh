cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['ms_per_step'], k['lstm_step_bwd'], k['lstm_step_fwd'])"; }
echo "dense bwd tiled"; FVTA_LSTM_WREG=1 run
echo "dense both PF8"; run
echo "ragged PF8 6144"; run --variant ragged
echo "ragged PF12 6144"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_wreg_bwd_abl12.so run --variant ragged
echo "dense PF12 all rows"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_wbwd_all12.so run
