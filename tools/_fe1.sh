cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['ms_per_step'], k['lstm_step_bwd'], k['lstm_step_fwd'])"; }
echo "dense dx wide"; run
echo "dense dx narrow"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_dxn.so run
echo "dense dx wide"; run
echo "dense dx narrow"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_dxn.so run
echo "ragged dx wide"; run --variant ragged
echo "ragged dx narrow"; FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_dxn.so run --variant ragged
