cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
show() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('train_l_fp8',{}); print('$1 head', d['ms_per_step'], 'fp8', r.get('ms_per_step'), 'bf16', r.get('bf16_ms_per_step'), 'train', d.get('train',{}).get('ms_per_step'), 'bs64', d.get('train_bs64',{}).get('ms_per_step'), 'frame', d.get('frame',{}).get('ms_per_step'), 'v52', d.get('v52',{}).get('ms_per_step'), 'bs1', d.get('infer_bs1',{}).get('ms_per_step'))"; }
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show full
timeout 2400 python -m pytest tests/test_gpu_training.py tests/test_gpu_dp_tuning.py tests/test_gpu_fp8.py -q -p no:cacheprovider > gpurun_out/r4a/tl.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r4a/tl.log | tail -8
