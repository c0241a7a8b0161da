#!/bin/bash
# step-by-step diagnosis of the ASan run (tools/asan_run.sh) on a GPU box
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
LIB=$(pwd)/yolopoint_amd/lib/ab/libASAN.so
echo "== 1 xnack only"; HSA_XNACK=1 timeout 120 python -c "import torch; print('cuda', torch.cuda.is_available()); x=torch.ones(4,device='cuda'); print(float(x.sum()))" 2>&1 | tail -3
echo "== 2 preload only"; ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT timeout 120 python -c "import torch; print('cuda', torch.cuda.is_available()); x=torch.ones(4,device='cuda'); print(float(x.sum()))" 2>&1 | tail -5
echo "== 3 preload + xnack + lib, one kernel"; HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT YP_HIP_LIB=$LIB timeout 300 python -c "
import torch, sys
sys.path.insert(0, 'tests')
from yolopoint_amd import _hip
print('lib', _hip.lib())
import numpy as np
from yolopoint_amd.utils import utils as U
from helpers import planted_heatmap
heat = planted_heatmap(64, 64, 20, 1)
print('pts', U.getPtsFromHeatmap(heat, 0.05, 4).shape)
" 2>&1 | tail -15
echo "== 4 rocminfo xnack"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -i -m3 "xnack\|gfx950"
