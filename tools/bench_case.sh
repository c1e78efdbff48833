#!/bin/bash
# one bench.py case, compact: tools/bench_case.sh <bench args...>   (prints wall ms, event median, bracketed kernels, fracs)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
python bench.py --also off --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['evidence']['headline']
print('$*', d['ms_per_step'], d['ms_per_step_event_median'], {k:round(x,3) for k,x in e['kernel_ms_per_step'].items()}, {k:[v['frac'],v['avg_launch_ms']] for k,v in e['rooflines'].items()})"
