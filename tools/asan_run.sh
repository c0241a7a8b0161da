#!/bin/bash
# Run GPU tests against the AddressSanitizer build of the library (make -C yolopoint_amd/csrc asan): host code and -- for every source but
# conv_igemm.hip, see the Makefile -- device code instrumented.  Through gpurun from the repo root:
#   tools/asan_run.sh <tag> [pytest arguments ...]      ->  gpurun_out/asan_<tag>.txt
# HSA_XNACK=1: the instrumented kernels are xnack+ code objects (shadow-memory faults are retried).  The Python interpreter is not an ASan
# binary, so the runtime is preloaded; leak checking is off (the interpreter and PyTorch "leak" by design).
set -u
TAG=${1:-run}
shift || true
ROOT=$(pwd)
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
LIB=$ROOT/yolopoint_amd/lib/ab/libASAN.so
[ -f "$LIB" ] || { echo "missing $LIB: make -C yolopoint_amd/csrc asan"; exit 2; }
mkdir -p gpurun_out
ARGS=${@:-tests/test_gpu_postproc.py tests/test_gpu_blocks.py -x -q}
HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0 LD_PRELOAD=$RT YP_HIP_LIB=$LIB \
  timeout 1500 python -m pytest $ARGS -p no:cacheprovider > gpurun_out/asan_$TAG.txt 2>&1
echo "rc=$?" >> gpurun_out/asan_$TAG.txt
grep -E "passed|failed|ERROR: AddressSanitizer|SUMMARY: AddressSanitizer|rc=" gpurun_out/asan_$TAG.txt | tail -12
