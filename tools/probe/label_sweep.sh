# same-box sweep of the label / loss-lane knobs after the CSR bucket sort got 4 x faster (round 6)
P="import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
run() { echo -n "$1 | $2: "; env $1 python bench.py --mode train $2 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for cfg in "--steps 30 --warmup 5" "--batch 64 --steps 6 --warmup 2" "--version l --batch 16 --dtype fp8 --steps 8 --warmup 3"; do
  for e in "YP_SIDE_WGS=256" "YP_LABELS_ORDER=first" "YP_LABELS_ORDER=after" "YP_SIDE_WGS=128" "YP_SIDE_WGS=512" "YP_SIDE_WGS=1024" "YP_SIDE_WGS=256"; do
    run "$e" "$cfg"
  done
done
