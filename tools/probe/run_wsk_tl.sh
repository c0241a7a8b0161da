cd $GRAFT_REPO_ROOT
export YP_HIP_LIB=$PWD/yolopoint_amd/lib/ab/libPW.so
for p in 0 1 2 7; do echo "probe $p"; YP_WSK_PROBE=$p python tools/probe/wsk_timeline.py --tile 74 --waves 0 2>&1 | grep -v amdgpu.ids | grep "inside\|iter 3 "; done
