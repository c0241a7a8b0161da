#!/bin/bash
# tools/probe/ab_infer.sh "ENV1" "ENV2" ...  : configs[1] forward under each environment, two rounds
for rep in 1 2; do
for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --only none --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$e', d['ms_per_step'], d['roofline']['frac'], d['roofline']['serial_launch_sum']['conv_us_per_step'])"
done
done
