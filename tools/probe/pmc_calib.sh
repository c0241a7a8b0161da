# calibrate GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES against kernel durations (same pass: counters + kernel trace)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_calib; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p -o m -- python $ROOT/bench.py --no-cpu-baseline --only none --steps 5 --warmup 2 --no-graph > $OUT/log.txt 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
d = "gpurun_out/pmc_calib/p"
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
dur = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))}
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
rows = [(k, dur[k][0], dur[k][1], v) for k, v in cnt.items() if k in dur and "conv" in dur[k][1]]
rows = rows[-60:]
for k, ns, name, v in rows[:12]:
    print(f"{ns/1e3:7.1f} us  GUI_ACTIVE {v.get('GRBM_GUI_ACTIVE',0):10.0f} ({v.get('GRBM_GUI_ACTIVE',0)/ns:6.2f} /ns)  SQ_BUSY {v.get('SQ_BUSY_CYCLES',0):12.0f} ({v.get('SQ_BUSY_CYCLES',0)/ns:7.2f} /ns)  MFMA_BUSY {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0):12.0f} ({v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/ns:7.2f} /ns)  {name[:60]}")
PY
