# one rocprofv3 kernel trace of the default training bench, distilled into gpurun_out/prof_tmp/tmp_train_step_*.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_tmp; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --steps 5 --warmup 2 "$@" > $OUT/trace_train.log 2>&1
cd $ROOT
python tools/profile_collect.py tmp > /dev/null 2>&1
rm -rf $OUT/trace_train
head -70 $OUT/tmp_train_step_kernels.txt | cut -c1-150
