#!/bin/bash
# Per-launch HBM bytes of configs[1] (YOLOPoint-s, bs 8, 640x640, f16) from the PMC counters, beside the algorithmic bytes:
#   tools/pmc_layers.sh <tag>   ->  gpurun_out/prof_<tag>/<tag>_layers_traffic.txt  (+ conv_traffic.json from the two-lane plan)
# The quick subset of tools/profile_round.sh (two PMC passes over the one-lane eager plan + two over the timed two-lane plan).
set -u
R=${1:-pl}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --only none --steps 50 --warmup 10"
python bench.py --no-cpu-baseline --only none --steps 300 --warmup 30 --layers $OUT/layers.txt > $OUT/bench_long.json 2> $OUT/bench.err
YP_INFER_LANES=0 python bench.py --no-cpu-baseline --only none --no-graph --steps 100 --warmup 10 --layers $OUT/layers_1lane.txt > $OUT/bench_1lane.json 2>> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
python $ROOT/tools/infer_sequence.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 50 > $OUT/${R}_infer_sequence.txt 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $BENCH > $OUT/pmc_write.log 2>&1
YP_INFER_LANES=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch1 -o f -- $BENCH --no-graph > $OUT/pmc_fetch1.log 2>&1
YP_INFER_LANES=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write1 -o w -- $BENCH --no-graph > $OUT/pmc_write1.log 2>&1
YP_INFER_LANES=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- $BENCH --no-graph > $OUT/pmc_mfma.log 2>&1
cd $ROOT
python tools/profile_collect.py $R > $OUT/collect.log 2>&1
tail -5 $OUT/collect.log
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch1 $OUT/pmc_write1 $OUT/pmc_mfma
cat $OUT/${R}_layers_traffic.txt
