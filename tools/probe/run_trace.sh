cd $GRAFT_REPO_ROOT
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4a; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_inf -o t -- python $ROOT/bench.py --no-cpu-baseline --only none --steps 20 --warmup 5 > $OUT/trace_inf.log 2>&1
cd $ROOT
f=$(find $OUT/trace_inf -name "*kernel_trace.csv" | head -1)
python tools/infer_sequence.py $f > $OUT/infer_sequence.txt
cp $f $OUT/infer_trace.csv; rm -rf $OUT/trace_inf
cat $OUT/infer_sequence.txt
