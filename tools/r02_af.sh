#!/bin/bash
# backward recurrence with / without the host's lengths hint (FVTA_LSTM_BWD_HINT=0 ignores it): ragged and dense variant
cd "$GRAFT_REPO_ROOT"
run() { python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', d['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'], 'dw', k['lstm_dw'])"; }
for h in 0 1 0 1; do FVTA_LSTM_BWD_HINT=$h run "ragged hint=$h" "--variant ragged"; done
for h in 0 1; do FVTA_LSTM_BWD_HINT=$h run "dense hint=$h" ""; done
