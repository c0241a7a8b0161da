cd $GRAFT_REPO_ROOT
for m in "" 55555555 0000ffff 33333333 0f0f0f0f 77777777 00ffffff; do
for h in 2 4; do
YP_SIDE_CUMASK=$m python bench.py --no-cpu-baseline --only none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cumask', '$m' or 'none', d['ms_per_step'])"
done; done
