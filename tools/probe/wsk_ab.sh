#!/bin/bash
# does the wave-private split-K kernel (tiles 71-76) move configs[1] when the tuner may pick it?  hot and cold tuning; per-launch tables kept
ALL=1,2,3,4,5,21,22,23,24,25,26,27,10,11,12,13,14,15,16,17,18,19,41,42,43,44,57,58
WSK=$ALL,71,72,73,74,75,76
B="python bench.py --no-cpu-baseline --only none --steps 400 --warmup 40"
pr='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["serial_launch_sum"]["conv_us_per_step"])'
for i in 1 2; do
  echo "default      : $(YP_TUNE_ONLY=$ALL $B --layers gpurun_out/wsk_layers_def.txt 2>/dev/null | python -c "$pr")"
  echo "with wsk     : $(YP_TUNE_ONLY=$WSK $B --layers gpurun_out/wsk_layers_wsk.txt 2>/dev/null | python -c "$pr")"
  echo "wsk, cold 64 : $(YP_TUNE_ONLY=$WSK YP_TUNE_COLD=64 $B --layers gpurun_out/wsk_layers_wskcold.txt 2>/dev/null | python -c "$pr")"
  echo "default cold : $(YP_TUNE_ONLY=$ALL YP_TUNE_COLD=64 $B --layers gpurun_out/wsk_layers_defcold.txt 2>/dev/null | python -c "$pr")"
done
paste <(awk '{print $1, $6}' gpurun_out/wsk_layers_def.txt) <(awk '{print $6}' gpurun_out/wsk_layers_wsk.txt) <(awk '{print $6}' gpurun_out/wsk_layers_wskcold.txt) <(awk '{print $6}' gpurun_out/wsk_layers_defcold.txt) | column -t | head -60
YP_TUNE_ONLY=$WSK YP_TUNE_DEBUG=1 $B 2>&1 | grep "^\[tune\]" | awk '{k=$2; if (!(k in best) || $5 < bestv[k]) {best[k]=$4; bestv[k]=$5}} END {for (k in best) print k, best[k], bestv[k]}' | sort | head -60
