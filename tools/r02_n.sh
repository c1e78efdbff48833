#!/bin/bash
# photo cell (64 sequences): backward step on 64-row block tiles (FVTA_LSTM_SMALL_ROWS) x c_t rebuilt (FVTA_LSTM_BWD_RC)
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_SMALL_ROWS=1 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_SMALL_ROWS=0 FVTA_LSTM_BWD_RC=0 run "small=0 rc=0"
FVTA_LSTM_SMALL_ROWS=1 FVTA_LSTM_BWD_RC=0 run "small=1 rc=0"
FVTA_LSTM_SMALL_ROWS=1 FVTA_LSTM_BWD_RC=1 run "small=1 rc=1"
FVTA_LSTM_SMALL_ROWS=0 FVTA_LSTM_BWD_RC=0 run "small=0 rc=0"
FVTA_LSTM_SMALL_ROWS=1 FVTA_LSTM_BWD_RC=0 run "small=1 rc=0"
FVTA_LSTM_SMALL_ROWS=1 FVTA_LSTM_BWD_RC=1 run "small=1 rc=1"
