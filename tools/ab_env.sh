#!/bin/bash
# same-box A/B of the headline under an environment switch: tools/ab_env.sh VAR VALUE_A VALUE_B [rounds] [extra bench args]
# alternating fresh processes; prints wall ms, event median and the bracketed kernels per run
cd "${GRAFT_REPO_ROOT:?}" || exit 1
var=$1; a=$2; b=$3; n=${4:-3}; shift 4
for i in $(seq $n); do
  for v in "$a" "$b"; do
    env $var=$v python bench.py --steps 20 --warmup 10 --also off --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['evidence']['headline']
print('$var=$v', d['ms_per_step'], d['ms_per_step_event_median'], {k:round(x,3) for k,x in e['kernel_ms_per_step'].items()})"
  done
done
