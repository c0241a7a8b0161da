#!/bin/bash
# rocprofv3 kernel trace of the training bench; prints per-step kernel time by kernel (plan construction diluted over 30 steps)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ttrace -o t -- python $R/bench.py --mode train --steps 20 --warmup 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections, re
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/ttrace/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# steady state: last 60% of dispatches by id
ids=sorted(int(r['Dispatch_Id']) for r in rows)
# find step boundaries via the Adam kernel? use time window: last 10 steps ~ take dispatches after the median start time
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t_end=int(rows[-1]['End_Timestamp'])
# take the final 10 steps: approx last 10*28ms = 280 ms
win=[r for r in rows if int(r['Start_Timestamp']) > t_end-280_000_000]
d=collections.defaultdict(lambda:[0,0])
for r in win:
    k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); k=re.sub(r'^void ','',k).split('(')[0][:80]
    d[k][0]+=1; d[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot=sum(v[1] for v in d.values())
print(f"window 280 ms (~10 steps): kernel time total {tot/1e6:.1f} ms -> {tot/1e6/10:.2f} ms/step busy")
for k,v in sorted(d.items(), key=lambda kv:-kv[1][1])[:40]:
    print(f"{k:82s} n/step={v[0]/10:7.1f} ms/step={v[1]/1e7:7.3f}")
PY
rm -rf $R/gpurun_out/ttrace
