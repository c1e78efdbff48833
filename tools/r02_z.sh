#!/bin/bash
# kernel trace of the ragged variant's train step: who ends the step (text recurrence + dW / dx, or the photo cell's chain)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_ragged_trace
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 bench.py --variant ragged --steps 3 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/err.log
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r02_ragged_trace/kt/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the last full step: split by adam_kernel occurrences
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
step=rows[a+1:b+1]
t0=int(step[0]['Start_Timestamp'])
def short(n): return n.split('(')[0].replace('void fvta::','').replace('fvta::','')[:46]
last={}
first={}
cnt=collections.Counter()
for r in step:
    k=(short(r['Kernel_Name']), r.get('Queue_Id') or r.get('Stream_Id'))
    s,e=(int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-t0)/1e6
    first.setdefault(k,s); last[k]=e; cnt[k]+=1
print('step length ms', (int(step[-1]['End_Timestamp'])-t0)/1e6)
for k in sorted(first,key=lambda k:first[k]):
    print('%-48s q%-3s n=%3d  first %.3f  last end %.3f' % (k[0],k[1],cnt[k],first[k],last[k]))
PY
