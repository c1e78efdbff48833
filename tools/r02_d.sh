#!/bin/bash
# round-2 set D: half-size tail tiles of the forward step kernel
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_FWD_TAIL=2 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "n64 or metric" 2>&1 | tail -3
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_FWD_TAIL=0 run "tail=off"
run "tail=on"
FVTA_LSTM_FWD_TAIL=0 run "tail=off"
run "tail=on"
