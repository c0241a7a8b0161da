cd $GRAFT_REPO_ROOT
bash tools/probe/run_trace.sh 2>&1 | grep -E "stem_conv2|span" | head -2
for f in 0 1 0 1; do
YP_FUSE_STEM2=$f python bench.py --no-cpu-baseline --only none --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fuse2 $f', d['ms_per_step'], d['roofline']['frac'])"
done
