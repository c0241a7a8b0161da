#!/bin/bash
# tools/probe/ab_env.sh <version> <dtype> "ENV1" "ENV2" ...   (each ENV a space-separated list of VAR=value; two rounds)
V="$1"; DT="$2"; shift 2
for rep in 1 2; do
for e in "$@"; do
    env $e python bench.py --mode train --version $V --dtype $DT --no-cpu-baseline --train-steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$e -$V $DT', d.get('ms_per_step'), 'ms')"
done
done
