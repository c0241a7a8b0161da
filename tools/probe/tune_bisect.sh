# bisect a failing random kernel-variant mixture of tools/probe/pair_grad_errors.py (YP_TUNE_RANDOM=$1): which signature is needed?
S=${1:-3}
fails() { YP_TUNE_RANDOM_LIMIT="$1" YP_TUNE_RANDOM=$S timeout 300 python tools/probe/pair_grad_errors.py 2>&1 | grep "n bad" | grep -qv "bad: 0 "; }
lo=0; hi=80
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(( (lo + hi) / 2 ))
  if fails "$lo,$mid"; then hi=$mid; elif fails "$mid,$hi"; then lo=$mid; else echo "needs both halves of [$lo,$hi) at $mid"; break; fi
  echo "in [$lo,$hi)"
done
echo "RESULT [$lo,$hi)"
YP_TUNE_DEBUG=1 YP_TUNE_RANDOM_LIMIT="$lo,$hi" YP_TUNE_RANDOM=$S timeout 300 python tools/probe/pair_grad_errors.py 2>&1 | grep -E "tune-random" | sed -n "$((lo+1)),$((hi))p"
YP_TUNE_RANDOM_LIMIT="$lo,$hi" YP_TUNE_RANDOM=$S timeout 300 python tools/probe/pair_grad_errors.py 2>&1 | grep -E "^grad|n bad" | head -12
