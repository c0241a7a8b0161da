cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
for cfg in "YP_TRAIN_BWD_LANES=0" "YP_TRAIN_BWD_LANES=1" "YP_TRAIN_BWD_LANES=0" "YP_TRAIN_BWD_LANES=1"; do
env $cfg python bench.py --mode train 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$cfg', d['ms_per_step'])
except Exception as e: print('$cfg', 'ERR', l[:300])"
done
YP_TRAIN_BWD_LANES=1 python bench.py --mode train 2>&1 | tail -30 > gpurun_out/r4a/bl.log
timeout 2400 python -m pytest tests/test_gpu_training.py tests/test_gpu_dp_tuning.py tests/test_gpu_bench_shapes.py tests/test_gpu_accuracy_parity.py -q -p no:cacheprovider > gpurun_out/r4a/tl.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r4a/tl.log | tail -12
