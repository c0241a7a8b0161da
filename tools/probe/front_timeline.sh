#!/bin/bash
# phase timelines (probe build, -DYP_TIMELINE) of the configs[1] front kernels and of one P5 1x1 layer
export YP_HIP_LIB=yolopoint_amd/lib/ab/libT.so
echo "== fused bottleneck C=32 at 160x160 (Bottleneck1.m.0 without the C3 tail)"; python tools/probe/timeline.py 32 160
echo "== fused bottleneck C=64 at 80x80"; python tools/probe/timeline.py 64 80
echo "== fused bottleneck C=128 at 40x40"; python tools/probe/timeline.py 128 40
echo "== 1x1 256->256 at 20x20, tile 4 (64x64)"; python tools/probe/timeline_conv.py 256 256 1 20 4
echo "== 1x1 256->256 at 20x20, tile 24 (64x64, 2 k tiles per barrier)"; python tools/probe/timeline_conv.py 256 256 1 20 24
echo "== 1x1 512->512 at 20x20, tile 4"; python tools/probe/timeline_conv.py 512 512 1 20 4
echo "== 3x3 256->256 at 20x20, tile 26"; python tools/probe/timeline_conv.py 256 256 3 20 26
echo "== 3x3 s2 128->256 out 40x40 halo tile 12"; python tools/probe/timeline_halo.py 128 256 2 40 12
echo "== 3x3 s2 64->128 out 80x80 halo tile 12"; python tools/probe/timeline_halo.py 64 128 2 80 12
