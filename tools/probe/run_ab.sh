cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
for cfg in "YP_TRAIN_DESCA_EARLY=0" "YP_TRAIN_DESCA_EARLY=1" "YP_TRAIN_DESCA_EARLY=0" "YP_TRAIN_DESCA_EARLY=1"; do
env $cfg python bench.py --mode train 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$cfg', d['ms_per_step'])
except Exception as e: print('$cfg', 'ERR', l[:300])"
done
