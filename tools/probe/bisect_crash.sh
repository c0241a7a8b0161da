cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
for f in tests/test_gpu_*.py; do
  [ "$f" = "tests/test_gpu_training.py" ] && continue
  timeout 600 python -m pytest $f "tests/test_gpu_training.py::test_parallel_training_graphs_equal_the_eager_launch_order" -x -q -p no:cacheprovider > /tmp/o.log 2>&1
  echo "$f rc=$? $(grep -c 'Segmentation' /tmp/o.log) $(tail -1 /tmp/o.log | cut -c1-80)"
done
