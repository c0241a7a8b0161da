# elimination experiments on the wave-private split-K kernel (probe build: make -C yolopoint_amd/csrc probewsk)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_conv_tiles.py -x -q 2>&1 | tail -3
export YP_HIP_LIB=$PWD/yolopoint_amd/lib/ab/libPW.so
for shp in c256_256_k3_20 c256_256_k1_20 c1024_512_k1_20 c256_512_k3s2_20 c128_128_k3_40 c512_256_k1_20; do
for pr in 0 1 2 7; do
  echo "== $shp probe $pr"
  YP_WSK_PROBE=$pr python tools/conv_bench.py --set s8 --tiles 71,72,73,74,75,76 --only $shp --iters 50 2>&1 | grep "^c"
done
done > gpurun_out/r4a/wsk_probe2.txt 2>&1
cat gpurun_out/r4a/wsk_probe2.txt
