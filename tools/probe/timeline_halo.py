"""Phase timeline of one workgroup of the 3x3 halo kernel (probe build, -DYP_TIMELINE):
YP_HIP_LIB=yolopoint_amd/lib/ab/libT.so python tools/probe/timeline_halo.py Cin Cout stride Hout tile"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder
c1, c2, st, Ho, tile = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
pb = PlanBuilder(8, _hip.YP_F16, dev); pb.autotune = False
x = pb.new_buf(Ho * st, Ho * st, c1); x.t.normal_()
pb.conv(x.view(), torch.randn(c2, c1, 3, 3) * 0.05, torch.zeros(c2), 3, st, 1, _hip.YP_ACT_SILU, tile=tile)
plan = pb.finish()
for _ in range(20): plan.run()
torch.cuda.synchronize()
ms = plan.time(200)
l = _hip.lib(); l.yp_debug_timeline.argtypes = [C.c_void_p]
buf = (C.c_longlong * 64)()
plan.run(); torch.cuda.synchronize()
assert l.yp_debug_timeline(buf) == 0
t = list(buf)
print(f"halo conv {c1}->{c2} s{st} out {Ho}x{Ho} B=8 tile {tile}: {ms*1e3:.1f} us per launch (back-to-back)")
names = {0: "entry", 1: "setup done", 40: "tap loop done", 41: "epilogue stores issued", 50: "  (args pinned, tile decoded)", 51: "  (halo offsets)", 52: "  (filter offsets)"}
for i in range(30): names[2 + i] = f"step {i} (chunk {i//3} filter row {i%3}) landed"
prev = t[0]
for i in sorted(names, key=lambda i: t[i] if t[i] else 1 << 62):
    if t[i] and (i < 2 or i >= 40 or t[i] > t[1]):
        print(f"  {names[i]:40s} +{t[i]-prev:6d}  = {t[i]-t[0]:7d} clk"); prev = t[i]
