#!/bin/bash
# rocprofv3 kernel trace of the frame pipeline (bench.py --mode frame) -> per-frame kernel table of the last 50 frames
V=${1:-l}; S=${2:-1280}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_frame -o f -- python $GRAFT_REPO_ROOT/bench.py --mode frame --version $V --size $S --steps 100 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_frame.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, re, os
rows=sorted(csv.DictReader(open('gpurun_out/prof_frame/f_kernel_trace.csv')), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "mnn_select_kernel" in r["Kernel_Name"]]
win=rows[idx[-51]+1: idx[-1]+1]
agg=collections.defaultdict(list)
for r in win:
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"^void ","",n).split("(")[0][:90]
    agg[n].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
tot=sum(sum(v) for v in agg.values()); span=int(win[-1]["End_Timestamp"])-int(win[0]["Start_Timestamp"])
lines=[f"# frame pipeline, last 50 frames: busy {tot/50/1e3:.0f} us/frame, span {span/50/1e3:.0f} us/frame, kernels/frame {len(win)/50:.0f}"]
conv=sum(sum(v) for k,v in agg.items() if "conv" in k or "bottleneck" in k)/50/1e3
lines.append(f"{'convolution kernels (all)':70s} {conv:9.1f} us/frame")
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    if "conv" in k or "bottleneck" in k: continue
    lines.append(f"{k:70s} {len(v)/50:6.1f} {sum(v)/50/1e3:9.1f} us/frame  avg {sum(v)/len(v)/1e3:8.2f}")
open('gpurun_out/frame_kernels.txt','w').write("\n".join(lines[:40])+"\n")
print("\n".join(lines[:22]))
PY
python - <<'PY'
# the last complete frame in start order: start offset, duration, gap to the end of the latest-ending earlier kernel
import csv, re
rows=sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open('gpurun_out/prof_frame/f_kernel_trace.csv'))))
idx=[i for i,r in enumerate(rows) if "mnn_select_kernel" in r[2]]
win=rows[idx[-2]+1: idx[-1]+1]
t0=win[0][0]; latest=None; out=[f"# one frame, {len(win)} kernels, span {(win[-1][1]-t0)/1e3:.1f} us"]
for s,e,n in win:
    n=re.sub(r"\(anonymous namespace\)::","",n); n=re.sub(r"^void ","",n).split("(")[0][:60]
    gap=(s-latest)/1e3 if latest else 0.0
    if gap>3 or gap<-3 or "conv" not in n and "bottleneck" not in n:
        out.append(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} {gap:8.1f}  {n}")
    latest=e if latest is None else max(latest,e)
open('gpurun_out/frame_sequence.txt','w').write("\n".join(out)+"\n")
print("\n".join(out))
PY
