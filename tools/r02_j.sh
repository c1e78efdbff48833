#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'], d['kernel_ms_per_step'])"; }
run "prio=0" "--side-priority 0"
run "prio=-1" "--side-priority -1"
run "prio=0" "--side-priority 0"
run "prio=-1" "--side-priority -1"
