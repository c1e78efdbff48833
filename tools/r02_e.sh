#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_NT=3 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
for m in 0 1 2 3 0 3; do FVTA_LSTM_NT=$m run "nt=$m"; done
