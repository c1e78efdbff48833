#!/bin/bash
# round 5: does a GPU-utilisation sampler running beside bench.py (what the round driver does: BENCH_r04.json carries a
# `gpu_busy` block with 8 samples) produce the one-off host stall?  Three samplers, each polling twice a second.
out=gpurun_out/stall2; mkdir -p $out
ls /sys/class/drm/ > $out/drm.txt 2>&1; which rocm-smi amd-smi >> $out/drm.txt 2>&1
run_set() {   # $1 = tag
  for i in 1 2 3; do
    python bench.py --steps 20 --warmup 5 --also off --no-cpu-baseline > $out/head_$1_$i.json 2>/dev/null
    python bench.py --steps 10 --warmup 3 --also off --no-cpu-baseline --variant ragged > $out/ragged_$1_$i.json 2>/dev/null
  done
}
run_set none
( while true; do cat /sys/class/drm/card*/device/gpu_busy_percent > /dev/null 2>&1; sleep 0.2; done ) & S=$!
run_set sysfs
kill $S
( while true; do rocm-smi --showuse > /dev/null 2>&1; sleep 0.2; done ) & S=$!
run_set rocmsmi
kill $S
( while true; do rocm-smi -a > /dev/null 2>&1; sleep 0.2; done ) & S=$!
run_set rocmsmia
kill $S
( while true; do amd-smi metric > /dev/null 2>&1; sleep 0.2; done ) & S=$!
run_set amdsmi
kill $S
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/stall2/*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f,'ERR',e); continue
    hw=r.get('host_watch',{})
    print('%-24s wall %.3f med %.3f host %.3f stepmax %s hostmax %s csw %s/%s'%(os.path.basename(f),r['ms_per_step'],r['ms_per_step_event_median'],r['host_enqueue_ms_per_step'],r.get('step_ms'),r.get('host_step_ms'),hw.get('vol_ctx_switches'),hw.get('invol_ctx_switches')))
PY
