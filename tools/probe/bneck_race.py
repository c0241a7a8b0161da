"""Is the fused Bottleneck kernel deterministic in isolation?  C in (32, 64, 128), with / without the C3 tail, big grids."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder
dev = torch.device("cuda:0")
for Cc, H, post in ((32, 160, False), (32, 160, True), (64, 80, True), (32, 64, True), (32, 96, False)):
    B = 8
    pb = PlanBuilder(B, _hip.YP_F16, dev); pb.autotune = False
    x = pb.new_buf(H, H, Cc); x.t.normal_()
    u = pb.new_buf(H, H, Cc); u.t.normal_()
    g = torch.Generator().manual_seed(1)
    w1, w2 = torch.randn(Cc, Cc, 1, 1, generator=g) * 0.1, torch.randn(Cc, Cc, 3, 3, generator=g) * 0.05
    w3 = torch.randn(2 * Cc, 2 * Cc, 1, 1, generator=g) * 0.1
    extra = {"pre": (w1, torch.zeros(Cc), _hip.YP_ACT_SILU)}
    if post:
        out = pb.new_buf(H, H, 2 * Cc).view()
        extra["post"] = (w3, torch.zeros(2 * Cc), _hip.YP_ACT_SILU, u.view())
    else:
        out = pb.new_buf(H, H, Cc).view()
    pb.conv(x.view(), w2, torch.zeros(Cc), 3, 1, 1, _hip.YP_ACT_SILU, out=out, res=x.view(), extra=extra)
    plan = pb.finish()
    outs = []
    for i in range(12):
        plan.run(); torch.cuda.synchronize()
        outs.append(out.buf.t.clone())
    nd = [int((o != outs[0]).sum()) for o in outs[1:]]
    bad = (outs[1] != outs[0]).nonzero()[:4].tolist() if nd[0] else []
    print(f"C={Cc} H={H} post={post}: differing elements vs run 0: {nd} {bad}")
