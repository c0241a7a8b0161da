"""Phase timeline of one workgroup of the fused Bottleneck kernel (probe build of the library with -DYP_TIMELINE):
YP_HIP_LIB=yolopoint_amd/lib/ab/libT.so python tools/probe/timeline.py [C] [H]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
pb = PlanBuilder(8, _hip.YP_F16, dev); pb.autotune = False
x = pb.new_buf(H, H, Cc); x.t.normal_()
w1, w2 = torch.randn(Cc, Cc, 1, 1) * 0.05, torch.randn(Cc, Cc, 3, 3) * 0.02
pb.conv(x.view(), w2, torch.zeros(Cc), 3, 1, 1, _hip.YP_ACT_SILU, res=x.view(), tile=11 if Cc >= 64 else 10, extra={"pre": (w1, torch.zeros(Cc), _hip.YP_ACT_SILU)})
plan = pb.finish()
for _ in range(20): plan.run()
torch.cuda.synchronize()
ms = plan.time(200)
buf = (C.c_longlong * 64)()
l = _hip.lib()
l.yp_debug_timeline.argtypes = [C.c_void_p]
plan.run(); torch.cuda.synchronize()
assert l.yp_debug_timeline(buf) == 0
t = list(buf)
names = {0: "kernel entry", 1: "setup done (offsets, bias loads issued)", 8: "phase A MFMAs done (barrier)", 9: "hidden written", 40: "tap loop done", 41: "epilogue stores issued"}
for c in range(Cc // 32): names[2 + c] = f"phase A chunk {c} landed"
for st in range(3 * (Cc // 32)): names[10 + st] = f"filter row {st} landed"
print(f"fused bottleneck C={Cc} {H}x{H} B=8: {ms*1e3:.1f} us per launch (back-to-back)")
prev = t[0]
for i in sorted(names):
    if t[i]:
        print(f"  {names[i]:42s} +{(t[i]-prev):6d}  = {t[i]-t[0]:7d} clk")
        prev = t[i]
