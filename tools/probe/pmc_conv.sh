cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in c32_64_k3s2_160 c64_128_k3s2_80 c64_64_k1_160; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$shape -o a -- python $R/tools/conv_bench.py --only $shape --iters 20 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$shape -o b -- python $R/tools/conv_bench.py --only $shape --iters 20 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_*')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv' in r['Kernel_Name']:
                agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d))
    for k,v in agg.items():
        n=max(len(x) for x in v.values())
        if n<10: continue
        print('  ',k, 'n=',n)
        print('     ', '  '.join(f"{c}={sorted(x)[len(x)//2]:.4g}" for c,x in sorted(v.items())))
PY
