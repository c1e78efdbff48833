#!/bin/bash
# attention backward: non-temporal stores of d_hinfo (bit 0) / loads of h (bit 1): FVTA_ATTN_BWD_NT
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
for v in 0 1 2 3 0 1 3; do FVTA_ATTN_BWD_NT=$v run "attn_bwd_nt=$v"; done
