#!/bin/bash
# forward step kernel: staged (LDS-transposed) epilogue vs the transposed-accumulator direct epilogue (FVTA_LSTM_FWD_DIRECT)
cd "$GRAFT_REPO_ROOT"
for v in 1 2; do
  FVTA_LSTM_FWD_DIRECT=$v python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
done
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
FVTA_LSTM_FWD_DIRECT=0 run "direct=0"
FVTA_LSTM_FWD_DIRECT=1 run "direct=1"
FVTA_LSTM_FWD_DIRECT=2 run "direct=2"
FVTA_LSTM_FWD_DIRECT=0 run "direct=0"
FVTA_LSTM_FWD_DIRECT=1 run "direct=1"
FVTA_LSTM_FWD_DIRECT=2 run "direct=2"
