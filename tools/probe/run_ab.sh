cd $GRAFT_REPO_ROOT
for b in 1 2 4; do
for cfg in "YP_INFER_LANES=0" "YP_LANES_EAGER=0" "YP_LANES_EAGER=1"; do
env $cfg python bench.py --no-cpu-baseline --only none --batch $b 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch', $b, '$cfg', d['ms_per_step'], d['value'])"
done; done
