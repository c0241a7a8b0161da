set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r02f; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --version l --batch 16 --dtype ${1:-fp8} --steps 4 --warmup 2 > $OUT/trace_train.log 2>&1
cd $ROOT
python tools/profile_collect.py r02f > /dev/null 2>&1
rm -rf $OUT/trace_train
head -40 $OUT/r02f_train_step_kernels.txt
