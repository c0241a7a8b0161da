cd $GRAFT_REPO_ROOT
for cfg in "4 4" "3 4" "2 4" "4 5" "3 5" "5 4" "2 3"; do set -- $cfg
for i in 1 2; do
YP_KP_AT=$1 YP_DESC_AT=$2 python bench.py --no-cpu-baseline --only none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('kp_at $1 desc_at $2', d['ms_per_step'])"
done; done
