#!/bin/bash
# ragged variant: attention forward per kernel choice
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --variant ragged --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['kernel_ms_per_step']['attn_fwd_main'], d['roofline_attention']['frac'], d['roofline_attention'].get('algorithmic_bytes'))"; }
for v in 0 3 2 1; do FVTA_ATTN_WAVE16=$v run "wave16=$v"; done
