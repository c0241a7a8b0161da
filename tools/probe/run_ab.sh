cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests/test_gpu_training.py -q -p no:cacheprovider -k "lanes or parallel or determin" > gpurun_out/r4a/tl.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r4a/tl.log | tail -12
