"""Second-generation main loop (tile ids 21..27) against the first-generation tiles: bit-identical outputs (same k order, fp32
accumulation) and back-to-back launch time per layer shape.  python tools/probe/v2_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

dev = torch.device("cuda:0")
# (name, Cin, Cout, k, s, Hout, batch)
CASES = [("64->64 1x1 160", 64, 64, 1, 1, 160, 8), ("128->128 1x1 80", 128, 128, 1, 1, 80, 8), ("256->256 1x1 40", 256, 256, 1, 1, 40, 8),
         ("512->512 1x1 20", 512, 512, 1, 1, 20, 8), ("1024->512 1x1 20", 1024, 512, 1, 1, 20, 8), ("512->256 1x1 20", 512, 256, 1, 1, 20, 8),
         ("256->512 3x3s2 20", 256, 512, 3, 2, 20, 8), ("128->256 3x3s2 40", 128, 256, 3, 2, 40, 8), ("96->72 3x3 17 (ragged)", 96, 72, 3, 1, 17, 3),
         ("160->40 1x1 9 (ragged)", 160, 40, 1, 1, 9, 5)]
for name, c1, c2, k, s, Ho, B in CASES:
    Hi = Ho * s
    torch.manual_seed(1)
    w, b = torch.randn(c2, c1, k, k) * 0.05, torch.randn(c2) * 0.1
    xin0 = torch.randn(B, Hi, Hi, c1, device=dev).half()
    outs, line = {}, []
    for tile in (4, 2, 3, 24, 26, 31, 33):
        pb = PlanBuilder(B, _hip.YP_F16, dev); pb.autotune = False
        x = pb.new_buf(Hi, Hi, c1); x.t.copy_(xin0)
        try:
            o = pb.conv(x.view(), w, b, k, s, k // 2, _hip.YP_ACT_SILU, tile=tile)
        except _hip.YpError as e:
            line.append(f"{tile}: n/a"); continue
        plan = pb.finish()
        plan.run(); torch.cuda.synchronize()
        outs[tile] = o.buf.t[..., :c2].float().clone()
        ms = plan.time(100)
        same = torch.equal(outs[tile], outs[4])
        line.append(f"{tile}: {ms*1e3:5.1f}us{'' if same else ' d%.1e' % float((outs[tile]-outs[4]).abs().max() / outs[4].abs().max())}")
    print(f"{name:26s} " + "  ".join(line), flush=True)
