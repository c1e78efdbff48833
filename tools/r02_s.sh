#!/bin/bash
# attention forward: one wave per 16-row tile (attn_fwd_wave16, FVTA_ATTN_WAVE16) vs the eight-waves-per-tile kernel
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "wave16" 2>&1 | tail -3
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['kernel_ms_per_step']['attn_fwd_main'], d['roofline_attention']['frac'])"; }
for v in 0 2 3 0 2 3; do FVTA_ATTN_WAVE16=$v run "wave16=$v"; done
