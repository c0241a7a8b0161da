#!/bin/bash
# same-box A/B of configs[1]: the library at yolopoint_amd/lib/ab/libOLD.so against the current one, alternating; per-launch tables of both
B="python bench.py --no-cpu-baseline --only none --steps 400 --warmup 40"
for i in 1 2 3; do
  echo "old: $(YP_HIP_LIB=yolopoint_amd/lib/ab/libOLD.so $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["gpu_ms_per_step_events"])')"
  echo "new: $($B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["gpu_ms_per_step_events"])')"
done
YP_HIP_LIB=yolopoint_amd/lib/ab/libOLD.so $B --layers gpurun_out/layers_old.txt > /dev/null 2>&1
$B --layers gpurun_out/layers_new.txt > /dev/null 2>&1
head -12 gpurun_out/layers_old.txt; head -12 gpurun_out/layers_new.txt; tail -1 gpurun_out/layers_old.txt; tail -1 gpurun_out/layers_new.txt
