#!/bin/bash
# per-launch duration of the photo cell's backward step kernel, one wave vs four-wave split-K (rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 4; do
  out=gpurun_out/r02_sk$v; mkdir -p $out
  export FVTA_LSTM_SMALL_SK=$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/err.log
  python3 - $out <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/ks/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'lstm_bwd_fused' in r['Name'] or 'lstm_dw_bf16' in r['Name']: print(sys.argv[1][-3:], r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
