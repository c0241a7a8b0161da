#!/bin/bash
# duration of the batched filter packer inside a training step: tools/probe/pack_time.sh <version> <batch>
V=${1:-l}; B=${2:-16}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_pack; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $ROOT/bench.py --mode train --version $V --batch $B --steps 4 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
python - <<P
import csv,glob,collections
f=glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pack_weight" in r["Kernel_Name"]: d[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("$V", k, len(v), "launches, median us", sorted(v)[len(v)//2])
P
rm -rf $OUT/t
