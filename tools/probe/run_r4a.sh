set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_conv_tiles.py -x -q 2>&1 | tail -15 > gpurun_out/r4a/tiles_test.log
cat gpurun_out/r4a/tiles_test.log
timeout 600 python tools/conv_bench.py --set s8 --tiles 0,4,24,26,71,72,73 --min-cin 32 > gpurun_out/r4a/conv_bench_s8.txt 2>&1
cat gpurun_out/r4a/conv_bench_s8.txt
YP_TUNE_DEBUG=1 timeout 900 python bench.py --no-cpu-baseline --only none --layers gpurun_out/r4a/layers.txt > gpurun_out/r4a/bench.log 2>&1
tail -3 gpurun_out/r4a/bench.log | cut -c1-1500
cat gpurun_out/r4a/layers.txt
