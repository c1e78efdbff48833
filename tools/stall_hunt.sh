#!/bin/bash
# round 5: hunt for the one-off host stall in bench.py's timed region (driver run r04: 14.565 ms wall vs 12.684 ms
# event median).  Fresh box: the FIRST process is the driver's own command; then fresh processes of the three cases
# that showed it, five each, collector left alone / frozen.
out=gpurun_out/stall; mkdir -p $out
python bench.py --gpus 1 --steps 20 --warmup 5 --gc default > $out/first_default.json 2> $out/first_default.err
for mode in default freeze; do
  for i in 1 2 3 4 5; do
    python bench.py --steps 20 --warmup 5 --also off --no-cpu-baseline --gc $mode > $out/head_${mode}_$i.json 2>/dev/null
    python bench.py --steps 10 --warmup 3 --also off --no-cpu-baseline --gc $mode --variant ragged > $out/ragged_${mode}_$i.json 2>/dev/null
    python bench.py --steps 5 --warmup 2 --also off --no-cpu-baseline --gc $mode --time-warp 5 > $out/tw_${mode}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/stall/*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f,'ERR',e); continue
    def show(name,r):
        hw=r.get('host_watch',{})
        print('%-28s wall %.3f med %.3f host %.3f stepmax %s hostmax %s gc %s majflt %s minflt %s csw %s/%s'%(name,r['ms_per_step'],r['ms_per_step_event_median'],r['host_enqueue_ms_per_step'],r.get('step_ms'),r.get('host_step_ms'),[g for g in hw.get('gc_collections',[]) if g[0]==2 or g[1]>1.0],hw.get('major_faults'),hw.get('minor_faults'),hw.get('vol_ctx_switches'),hw.get('invol_ctx_switches')))
    show(os.path.basename(f),r)
    for k,v in (r.get('also') or {}).items():
        if 'error' in v: print('   ',k,v['error']); continue
        show('   '+k,v)
PY
