#!/bin/bash
# round-2 measurement set B: 256x256 / four-wave (128x128 wave tile) LSTM kernels, overlap variants
cd "$GRAFT_REPO_ROOT"
FVTA_LSTM_TILE128=15 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -4
FVTA_LSTM_OVERLAP=3 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k overlapped 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['kernel_ms_per_step'])"; }
for m in 0 1 2 4 8 15; do FVTA_LSTM_TILE128=$m run "tile128=$m"; done
FVTA_LSTM_OVERLAP=1 run "overlap=dx"
FVTA_LSTM_OVERLAP=1 FVTA_LSTM_TILE128=15 run "overlap=dx tile128=15"
