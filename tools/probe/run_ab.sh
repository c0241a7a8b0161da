cd $GRAFT_REPO_ROOT
OLD=1,2,3,4,5,21,22,23,24,25,26,27,10,11,12,13,14,15,41,42,43,44,57,58
for i in 1 2 3; do
YP_TUNE_ONLY=$OLD python bench.py --no-cpu-baseline --only none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('no wsk', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --no-cpu-baseline --only none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('wsk   ', d['ms_per_step'], d['roofline']['frac'], d['config']['launch'])"
done
python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
