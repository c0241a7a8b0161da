#!/bin/bash
# tools/probe/ab_v52.sh "ENV1" "ENV2" ...: the v52 record under each environment, two rounds
for rep in 1 2; do
for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --only v52 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); v=d['v52']; print('$e', 'v52', v['ms_per_step'], v['roofline']['frac'])"
done
done
