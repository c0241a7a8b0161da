# same-box sweep of the InfoNCE gather grid cap after the CSR bucket sort got 4 x faster (round 6)
P="import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
run() { echo -n "$1 | $2: "; env $1 python bench.py --mode train $2 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for e in "YP_NCE_WGS=256" "YP_NCE_WGS=384" "YP_NCE_WGS=512" "YP_NCE_WGS=768" "YP_NCE_WGS=192" "YP_NCE_WGS=256"; do run "$e" "--batch 64 --steps 6 --warmup 2"; done
for e in "YP_NCE_WGS=96" "YP_NCE_WGS=128" "YP_NCE_WGS=192" "YP_NCE_WGS=256" "YP_NCE_WGS=64" "YP_NCE_WGS=96"; do run "$e" "--version l --batch 16 --dtype fp8 --steps 8 --warmup 3"; done
for e in "YP_NCE_WGS=256" "YP_NCE_WGS=192" "YP_NCE_WGS=384" "YP_NCE_WGS=256"; do run "$e" "--steps 30 --warmup 5"; done
