#!/bin/bash
# photo cell's backward step with its k-loop split over four waves (FVTA_LSTM_SMALL_SK=4, default) vs one wave (0)
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
run() { python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', d['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'], 'dw', k['lstm_dw'])"; }
for v in 0 4 0 4; do FVTA_LSTM_SMALL_SK=$v run "ragged sk=$v" "--variant ragged"; done
for v in 0 4 0 4; do FVTA_LSTM_SMALL_SK=$v run "dense sk=$v" ""; done
