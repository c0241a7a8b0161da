#!/bin/bash
# Regenerates the evidence under profiles/ for one round on a GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1. the default bench line, 2. rocprofv3 --kernel-trace of the same command, 3. two separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE; never combined with other trace domains), 4. the training step line + its kernel trace.
# Writes everything to gpurun_out/prof_<round>/; tools/profile_collect.py then distils profiles/<round>_*.{txt,json}.
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --only none --steps 50 --warmup 10"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --no-cpu-baseline --only none --steps 300 --warmup 30 --layers $OUT/layers.txt > $OUT/bench_long.json 2>> $OUT/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
# one forward in start order (start offset, duration, overlap with the other lane): the two-lane schedule as it ran
python $ROOT/tools/infer_sequence.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 50 > $OUT/${R}_infer_sequence.txt 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $BENCH > $OUT/pmc_write.log 2>&1
# MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x 256 CUs); ONE-LANE eager replay so that the dispatch order is the plan order
YP_INFER_LANES=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- $BENCH --no-graph > $OUT/pmc_mfma.log 2>&1
# per-launch HBM bytes (the same one-lane eager plan; its own per-launch table gives the row names)
cd $ROOT
YP_INFER_LANES=0 python bench.py --no-cpu-baseline --only none --no-graph --steps 100 --warmup 10 --layers $OUT/layers_1lane.txt > $OUT/bench_1lane.json 2>> $OUT/bench_default.err
cd /tmp
YP_INFER_LANES=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch1 -o f -- $BENCH --no-graph > $OUT/pmc_fetch1.log 2>&1
YP_INFER_LANES=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write1 -o w -- $BENCH --no-graph > $OUT/pmc_write1.log 2>&1
cd $ROOT
python bench.py --mode train --steps 20 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --steps 5 --warmup 2 > $OUT/trace_train.log 2>&1
cd $ROOT
python tools/train_bench.py > $OUT/train_bench.log 2>&1
cp gpurun_out/train_fwd_ops.txt gpurun_out/train_bwd_ops.txt $OUT/ 2>/dev/null
cp $OUT/${R}_train_step_kernels.txt $OUT/keep_train_step_kernels.txt 2>/dev/null
# BASELINE configs[4] shape on one GPU: YOLOPoint-l, fp8 Conv operands (and the same step in bf16), + the kernel list of one fp8 step
python bench.py --mode train --version l --batch 16 --dtype fp8 --steps 8 --warmup 3 > $OUT/bench_train_l_fp8.json 2> $OUT/bench_train_l_fp8.err
python bench.py --mode train --version l --batch 16 --dtype bf16 --steps 8 --warmup 3 > $OUT/bench_train_l_bf16.json 2>> $OUT/bench_train_l_fp8.err
python bench.py --mode frame --version l --size 1280 > $OUT/bench_frame.json 2> $OUT/bench_frame.err
python bench.py --mode export > $OUT/bench_export.json 2> $OUT/bench_export.err
python tools/profile_collect.py $R
# one fp8 step of YOLOPoint-l in the same form (the collector distils whatever lies in trace_train)
mkdir -p $OUT/fp8 && mv $OUT/${R}_train_step_kernels.txt $OUT/${R}_train_step_sequence.txt $OUT/${R}_train_kernel_trace.txt $OUT/fp8/ 2>/dev/null
rm -rf $OUT/trace_train
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --version l --batch 16 --dtype fp8 --steps 4 --warmup 2 > $OUT/trace_train_fp8.log 2>&1
cd $ROOT
python tools/profile_collect.py $R > /dev/null 2>&1
mv $OUT/${R}_train_step_kernels.txt $OUT/${R}_train_l_fp8_step_kernels.txt
mv $OUT/fp8/* $OUT/ && rmdir $OUT/fp8
# ... and one step of the `train_bs64` record (ONE batch of 64 samples, YOLOPoint-s, bf16)
mkdir -p $OUT/keep && mv $OUT/${R}_train_step_kernels.txt $OUT/${R}_train_step_sequence.txt $OUT/${R}_train_kernel_trace.txt $OUT/keep/ 2>/dev/null
rm -rf $OUT/trace_train
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python $ROOT/bench.py --mode train --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace_train_bs64.log 2>&1
cd $ROOT
python tools/profile_collect.py $R > /dev/null 2>&1
mv $OUT/${R}_train_step_kernels.txt $OUT/${R}_train_bs64_step_kernels.txt
mv $OUT/keep/* $OUT/ && rmdir $OUT/keep
# HBM traffic of the training steps (roofline.traffic of the train / train_l_fp8 records): separate PMC passes
cd /tmp
for cfg in "s 8 bf16" "l 16 fp8"; do
  set -- $cfg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tf -o f -- python $ROOT/bench.py --mode train --version $1 --batch $2 --dtype $3 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_tf.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tw -o w -- python $ROOT/bench.py --mode train --version $1 --batch $2 --dtype $3 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_tw.log 2>&1
  python $ROOT/tools/train_traffic.py $OUT/pmc_tf $OUT/pmc_tw ${1}_${2}_${3} $OUT/train_traffic.json
  rm -rf $OUT/pmc_tf $OUT/pmc_tw
done
# ... and of one frame of the frame pipeline (configs[3]): the dispatches between two mnn_select_kernel launches
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tf -o f -- python $ROOT/bench.py --mode frame --version l --size 1280 --steps 30 --warmup 5 > $OUT/pmc_tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tw -o w -- python $ROOT/bench.py --mode frame --version l --size 1280 --steps 30 --warmup 5 > $OUT/pmc_tw.log 2>&1
python $ROOT/tools/train_traffic.py $OUT/pmc_tf $OUT/pmc_tw frame_l_1280_f16 $OUT/train_traffic.json mnn_select_kernel
rm -rf $OUT/pmc_tf $OUT/pmc_tw
# ... and of one forward of the v52 record (YOLOPointv52-s, bs 8, 640x640): the dispatches between its last two stem launches
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tf -o f -- python $ROOT/bench.py --no-cpu-baseline --only v52 --steps 5 --warmup 2 > $OUT/pmc_tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tw -o w -- python $ROOT/bench.py --no-cpu-baseline --only v52 --steps 5 --warmup 2 > $OUT/pmc_tw.log 2>&1
python $ROOT/tools/train_traffic.py $OUT/pmc_tf $OUT/pmc_tw v52_s_8_f16 $OUT/train_traffic.json stem_conv
rm -rf $OUT/pmc_tf $OUT/pmc_tw
cd $ROOT
# the 8-wave kernels: per-layer tables (16-bit and 8-bit) and the SQ / LDS counters of one deep layer per schedule
python tools/conv_bench.py --set l32 --dtype bf16 --tiles 0,3,41,42,43,44,57,58,61,62 --min-cin 64 --iters 50 > $OUT/${R}_conv_bench_l32_bf16.txt 2>/dev/null
python tools/conv_bench.py --set s64 --dtype f16 --tiles 0,3,41,42,43,44,57,58 --min-cin 64 > $OUT/${R}_conv_bench_s64_f16.txt 2>/dev/null
python tools/conv_bench_fp8.py --tiles 0,2,3,57 > $OUT/${R}_conv_bench_fp8_l32.txt 2>/dev/null
bash tools/probe/pmc_mma8.sh l32 c256_256_k3_40 41,57,58,62,3 > $OUT/${R}_mma8_pmc_c256_256_k3_40.txt 2>&1
bash tools/probe/pmc_mma8.sh l32 c2048_1024_k1_20 41,57,3 > $OUT/${R}_mma8_pmc_c2048_1024_k1_20.txt 2>&1
rm -rf $ROOT/gpurun_out/pmc8_*
# raw traces are large: keep only what profile_collect distilled
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch1 $OUT/pmc_write1 $OUT/pmc_mfma $OUT/trace_train
ls -la $OUT
