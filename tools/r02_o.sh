#!/bin/bash
# the side-stream overlap modes again, now that the photo cell's backward chain is short (FVTA_LSTM_OVERLAP bits, lstm.hip)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_OVERLAP=0 run "ov=0"
FVTA_LSTM_OVERLAP=1 run "ov=1 dx per group on the side stream"
FVTA_LSTM_OVERLAP=2 run "ov=2 dW per group on the side stream"
FVTA_LSTM_OVERLAP=8 run "ov=8 dx beside dW"
FVTA_LSTM_OVERLAP=0 run "ov=0"
