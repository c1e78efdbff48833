#!/bin/bash
# dx: atomics in one launch vs one launch per direction (store, then add): FVTA_LSTM_DX_2PASS
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_DX_2PASS=1 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_DX_2PASS=0 run "2pass=0"
FVTA_LSTM_DX_2PASS=1 run "2pass=1"
FVTA_LSTM_DX_2PASS=0 run "2pass=0"
FVTA_LSTM_DX_2PASS=1 run "2pass=1"
