# which pair (i in 0..4, j in 5..9) of the picks of the failing mixture reproduces the failure?
P=(5 3 4 5 4 3 2 5 2 4)
for i in 0 1 2 3 4; do for j in 5 6 7 8 9; do
  r=$(YP_TUNE_RANDOM=1 YP_TUNE_FORCE="$i:${P[$i]},$j:${P[$j]}" timeout 300 python tools/probe/pair_grad_errors.py 2>&1 | grep "n bad")
  echo "sig $i=${P[$i]} sig $j=${P[$j]}: $r"
done; done
