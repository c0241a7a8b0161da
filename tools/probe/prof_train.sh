set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r02a; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --steps 5 --warmup 2 > $OUT/trace_train.log 2>&1
cd $ROOT
python tools/profile_collect.py r02a > /dev/null 2>&1
rm -rf $OUT/trace_train
cat $OUT/r02a_train_step_kernels.txt
