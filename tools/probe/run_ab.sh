cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
YP_SIDE_PRIORITY=0 python bench.py --no-cpu-baseline --only none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('side default prio', d['ms_per_step'])"
python bench.py --no-cpu-baseline --only none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('side lowest prio ', d['ms_per_step'])"
done
