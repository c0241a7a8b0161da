"""Phase timeline of one workgroup of the generic implicit-GEMM conv kernel (probe build, -DYP_TIMELINE):
YP_HIP_LIB=yolopoint_amd/lib/ab/libT.so python tools/probe/timeline_conv.py Cin Cout k H tile"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder
c1, c2, k, H, tile = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
pb = PlanBuilder(8, _hip.YP_F16, dev); pb.autotune = False
x = pb.new_buf(H, H, c1); x.t.normal_()
pb.conv(x.view(), torch.randn(c2, c1, k, k) * 0.05, torch.zeros(c2), k, 1, k // 2, _hip.YP_ACT_SILU, tile=tile)
plan = pb.finish()
for _ in range(20): plan.run()
torch.cuda.synchronize()
ms = plan.time(200)
l = _hip.lib(); l.yp_debug_timeline.argtypes = [C.c_void_p]
buf = (C.c_longlong * 64)()
plan.run(); torch.cuda.synchronize()
assert l.yp_debug_timeline(buf) == 0
t = list(buf)
print(f"conv {c1}->{c2} k{k} {H}x{H} B=8 tile {tile}: {ms*1e3:.1f} us per launch (back-to-back)")
names = {0: "entry", 1: "setup done, bias loads issued", 40: "k loop done", 41: "epilogue stores issued", 50: "  (tile decoded: first arguments in)", 51: "  (pixel rows decoded)", 52: "  (filter offsets, arguments pinned)"}
for kt in range(24): names[2 + kt] = f"k tile {kt} landed"
prev = t[0]
for i in sorted(names, key=lambda i: t[i] if t[i] else 1 << 62):
    if t[i] and (i < 2 or i >= 40 or t[i] > t[1]):
        print(f"  {names[i]:34s} +{t[i]-prev:6d}  = {t[i]-t[0]:7d} clk"); prev = t[i]
