#!/bin/bash
# The `statistical` tests (training curves, gradient statistics, trained-checkpoint parity) under the autotuner's picks and under random
# kernel-variant mixtures (YP_TUNE_RANDOM seeds): one line per run -> gpurun_out/stat_seeds_<tag>.log
TAG=${1:-a}
OUT=gpurun_out/stat_seeds_$TAG.log
: > $OUT
for seed in tuned 1 2 3 4 5 6 7 8 9; do
  if [ $seed = tuned ]; then unset YP_TUNE_RANDOM; else export YP_TUNE_RANDOM=$seed; fi
  python -m pytest tests -m "gpu and statistical" -q -s > gpurun_out/stat_seed_$seed.log 2>&1
  echo "seed $seed: $(tail -1 gpurun_out/stat_seed_$seed.log)" >> $OUT
  grep -h "per-step deviation\|map50\|tensors keep" gpurun_out/stat_seed_$seed.log >> $OUT
done
cat $OUT
