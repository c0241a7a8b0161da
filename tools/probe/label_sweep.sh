# same-box sweeps of the label / loss-lane knobs after the CSR bucket sort got 4 x faster (round 6).
#   bash tools/probe/label_sweep.sh side   -> YP_SIDE_WGS / YP_LABELS_ORDER  (measured: +-1 %, the defaults stay)
#   bash tools/probe/label_sweep.sh nce    -> YP_NCE_WGS (the InfoNCE gathers' grid cap beside the YOLO-branch backward)
P="import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
run() { echo -n "$1 | $2: "; env $1 python bench.py --mode train $2 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
S8="--steps 30 --warmup 5"; S64="--batch 64 --steps 6 --warmup 2"; L16="--version l --batch 16 --dtype fp8 --steps 8 --warmup 3"
if [ "${1:-side}" = "side" ]; then
  for cfg in "$S8" "$S64" "$L16"; do
    for e in "YP_SIDE_WGS=256" "YP_LABELS_ORDER=first" "YP_LABELS_ORDER=after" "YP_SIDE_WGS=128" "YP_SIDE_WGS=512" "YP_SIDE_WGS=1024" "YP_SIDE_WGS=256"; do run "$e" "$cfg"; done
  done
else
  for e in "YP_NCE_WGS=256" "YP_NCE_WGS=384" "YP_NCE_WGS=512" "YP_NCE_WGS=768" "YP_NCE_WGS=192" "YP_NCE_WGS=256"; do run "$e" "$S64"; done
  for e in "YP_NCE_WGS=96" "YP_NCE_WGS=128" "YP_NCE_WGS=192" "YP_NCE_WGS=256" "YP_NCE_WGS=64" "YP_NCE_WGS=96"; do run "$e" "$L16"; done
  for e in "YP_NCE_WGS=256" "YP_NCE_WGS=192" "YP_NCE_WGS=384" "YP_NCE_WGS=256"; do run "$e" "$S8"; done
fi
