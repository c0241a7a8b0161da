cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
show() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['train_l_fp8']; print('$1 fp8', r['ms_per_step'], 'bf16', r.get('bf16_ms_per_step'), 'frame', d.get('frame',{}).get('ms_per_step'))"; }
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show full
python bench.py --no-cpu-baseline --only fp8 2>/dev/null | tail -1 | show only_fp8
python bench.py --no-cpu-baseline --only train,fp8 2>/dev/null | tail -1 | show train_fp8
python bench.py --no-cpu-baseline --only train64,fp8 2>/dev/null | tail -1 | show train64_fp8
