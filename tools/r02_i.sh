#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_OVERLAP=4 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k overlapped 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_OVERLAP=0 run "ov=0"
FVTA_LSTM_OVERLAP=4 run "ov=4 delay 0"
FVTA_LSTM_OVERLAP=$((4 + (40<<8))) run "ov=4 delay 40us"
FVTA_LSTM_OVERLAP=$((4 + (75<<8))) run "ov=4 delay 75us"
FVTA_LSTM_OVERLAP=$((4 + (110<<8))) run "ov=4 delay 110us"
FVTA_LSTM_OVERLAP=0 run "ov=0"
