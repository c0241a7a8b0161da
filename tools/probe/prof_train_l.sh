# one steady-state YOLOPoint-l training step, per-kernel: tools/probe/prof_train_l.sh <dtype> <tag>
set -u
DT=${1:-fp8}; TAG=${2:-r03l}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --version l --batch 16 --dtype $DT --steps 4 --warmup 2 > $OUT/trace_train.log 2>&1
cd $ROOT
python tools/profile_collect.py $TAG > /dev/null 2>&1
rm -rf $OUT/trace_train
head -45 $OUT/${TAG}_train_step_kernels.txt
