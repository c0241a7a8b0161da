#!/bin/bash
# tools/probe/ab_env_b.sh <version> <dtype> <batch> "ENV1" "ENV2" ...
V="$1"; DT="$2"; B="$3"; shift 3
for rep in 1 2; do
for e in "$@"; do
    env $e python bench.py --mode train --version $V --dtype $DT --batch $B --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$e -$V $DT b$B', d.get('ms_per_step'), 'ms')"
done
done
