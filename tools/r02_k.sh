#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_OVERLAP=8 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k overlapped 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_OVERLAP=0 run "ov=0"
FVTA_LSTM_OVERLAP=8 run "ov=8 dx||dW"
FVTA_LSTM_OVERLAP=0 run "ov=0"
FVTA_LSTM_OVERLAP=8 run "ov=8 dx||dW"
