# SQ / LDS counters of one conv_bench shape per kernel variant: tools/probe/pmc_mma8.sh <set> <shape> <tiles>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SET=${1:-l32}; SHAPE=${2:-c256_256_k3_40}; TILES=${3:-41}
OUT=$R/gpurun_out/pmc8_${SHAPE}
rm -rf $OUT
CMD="python $R/tools/conv_bench.py --set $SET --dtype bf16 --only $SHAPE --tiles $TILES --iters 10"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT -o b -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections, os
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    n=max(len(x) for x in v.values())
    if n<5: continue
    print(k, 'n=',n)
    for c,x in sorted(v.items()):
        print(f"     {c:36s} {sorted(x)[len(x)//2]:.5g}")
PY
