#!/bin/bash
# A/B of the training records under an environment switch: tools/probe/ab_train.sh "YP_NCE_GEN=1" "YP_NCE_GEN=2" [s|l|both]
A="$1"; B="$2"; W="${3:-both}"
run() {   # env, version, dtype
    env $1 python bench.py --mode train --version $2 --dtype $3 --no-cpu-baseline --train-steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 -$2 $3', d.get('ms_per_step'), 'ms')"
}
for rep in 1 2; do
  if [ "$W" != "l" ]; then run "$A" s bf16; run "$B" s bf16; fi
  if [ "$W" != "s" ]; then run "$A" l bf16; run "$B" l bf16; fi
done
