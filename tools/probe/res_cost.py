"""Cost of the residual (accumulate-into-output) epilogue path: the same convolution with and without `res = out`, per kernel family.
python tools/probe/res_cost.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

dev = torch.device("cuda:0")
B = 32
SHAPES = [(256, 128, 1, 80), (128, 128, 3, 80), (512, 256, 1, 40), (256, 256, 3, 40), (128, 64, 1, 160), (1024, 512, 1, 20)]
for c1, c2, k, H in SHAPES:
    row = []
    for res in (False, True):
        pb = PlanBuilder(B, _hip.YP_BF16, dev)
        x = pb.new_buf(H, H, c1); x.t.normal_()
        o = pb.new_buf(H, H, c2); o.t.normal_()
        w = torch.randn(c2, c1, k, k) * 0.02
        pb.conv(x.view(), w, None, k, 1, k // 2, _hip.YP_ACT_NONE, out=o.view(), res=o.view() if res else None)
        plan = pb.finish()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                plan.run()
            row.append(plan.time(50) * 1e3)
    print(f"c{c1}->{c2} k{k} @{H}x{H}x{B}: plain {row[0]:7.1f} us   accumulate {row[1]:7.1f} us   (+{(row[1] / row[0] - 1) * 100:.0f} %)", flush=True)
