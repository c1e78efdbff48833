#!/bin/bash
# attention forward: attn_fwd_pair16 with LDS-address-space flag accesses (no vmcnt(0) in the polls) vs attn_fwd_rows16
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "attention or attn or wave16" 2>&1 | tail -3
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['kernel_ms_per_step']['attn_fwd_main'], d['roofline_attention']['frac'])"; }
for v in 0 3 2 0 3 2; do FVTA_ATTN_WAVE16=$v run "wave16=$v"; done
python tools/stress_attn.py 2>&1 | tail -2
