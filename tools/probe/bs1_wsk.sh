B="python bench.py --no-cpu-baseline --only none --steps 400 --warmup 50"
BASE=1,2,3,4,5,21,22,23,24,25,26,27,10,11,12,13,14,15,16,17,18,19,41,42,43,44,57,58,61,62
for b in 1 8; do
  $B --batch $b --layers gpurun_out/r6_layers_bs${b}.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prod bs$b', d['ms_per_step'])"
  YP_HIP_LIB=yolopoint_amd/lib/ab/libPW.so YP_TUNE_ONLY=$BASE,71,72,73,74,75,76 $B --batch $b --layers gpurun_out/r6_layers_bs${b}_wsk.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wsk  bs$b', d['ms_per_step'])"
  YP_HIP_LIB=yolopoint_amd/lib/ab/libPW.so $B --batch $b | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('libPW-no-wsk bs$b', d['ms_per_step'])"
done
cat gpurun_out/r6_layers_bs1.txt
grep -n "conv" gpurun_out/r6_layers_bs1_wsk.txt | awk '{print $1, $6}' | head -60
