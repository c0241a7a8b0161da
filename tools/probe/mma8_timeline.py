"""Segment clocks of the 8-wave convolution kernel (probe build lib/ab/libP8.so): waves 0 (group 0) and 4 (group 1) of workgroup 0, k tile 6.
YP_HIP_LIB=.../libP8.so python tools/probe/mma8_timeline.py <tile> [shape]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 41
c1, c2, k, Ho, B = 256, 256, 3, 40, 32
dev = torch.device("cuda:0")
pb = PlanBuilder(B, _hip.YP_BF16, dev)
x = pb.new_buf(Ho, Ho, c1); x.t.normal_()
pb.conv(x.view(), torch.randn(c2, c1, k, k) * 0.02, torch.zeros(c2), k, 1, k // 2, _hip.YP_ACT_SILU if os.environ.get('YP_TL_ACT', '1') == '1' else _hip.YP_ACT_NONE, tile=tile)
plan = pb.finish()
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
l = _hip.lib()
l.yp_debug_mma8_timeline.argtypes = [C.POINTER(C.c_ulonglong)]
assert l.yp_debug_mma8_timeline(buf) == 0
base = min(buf[w * 32] for w in range(8) if buf[w * 32])
for w in range(8):
    ts = [buf[w * 32 + i] for i in range(24)]
    print(f"tile {tile} wave {w}: " + " ".join(f"{(t - base) if t else -1:5d}" for t in ts))
    xs = [buf[w * 32 + 24 + i] for i in range(6)]
    print(f"   kernel phases (clk): entry->setup {xs[1]-xs[0]}, prologue DMA {xs[2]-xs[1]}, k loop {xs[3]-xs[2]}, stats/bias {xs[4]-xs[3]}, stores {xs[5]-xs[4]}, total {xs[5]-xs[0]}")
