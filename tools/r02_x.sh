#!/bin/bash
# attn_fwd_pair16 prologue: question staging with every load issued before the first LDS write; metric + ragged lines
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "attention or attn or wave16" 2>&1 | tail -2
run() { python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['kernel_ms_per_step']['attn_fwd_main'], d['roofline_attention']['frac'])"; }
for v in 0 3 0 3; do FVTA_ATTN_WAVE16=$v run "metric wave16=$v" ""; done
for v in 0 3; do FVTA_ATTN_WAVE16=$v run "ragged wave16=$v" "--variant ragged"; done
python tools/bench_attn.py 2>&1 | grep "attn fwd"
