#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FVTA_GLDS_SP=15 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
for m in 0 1 2 4 7 0 7; do FVTA_GLDS_SP=$m run "sp=$m"; done
