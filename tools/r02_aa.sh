#!/bin/bash
# backward step tile choice (256 x 256 default vs 256 x 128, FVTA_LSTM_BWD_NARROW_TILE=1) on the dense and the ragged variant
cd "$GRAFT_REPO_ROOT"
run() { python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', d['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'], 'dw', k['lstm_dw'])"; }
for nt in 0 1 0 1; do FVTA_LSTM_BWD_NARROW_TILE=$nt run "ragged narrow=$nt" "--variant ragged"; done
for nt in 0 1; do FVTA_LSTM_BWD_NARROW_TILE=$nt run "dense narrow=$nt" ""; done
