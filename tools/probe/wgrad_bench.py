"""Time yp_conv_wgrad on the YOLOPoint-s training shapes (bs 8, 640): us, TFLOP/s.  Env knobs (probe only):
YP_WG_CAP=<n> pixel-split cap, YP_WG_NOSTORE=1 skip the final atomics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, lib, check

SHAPES = [  # Cin, Cout, k, stride, Hout
    (32, 64, 3, 2, 160), (32, 32, 3, 1, 160), (64, 32, 1, 1, 160), (32, 32, 1, 1, 160), (64, 64, 1, 1, 160),
    (64, 128, 3, 2, 80), (64, 64, 3, 1, 80), (128, 64, 1, 1, 80), (64, 64, 1, 1, 80), (128, 128, 1, 1, 80), (128, 128, 3, 1, 80), (256, 64, 1, 1, 80),
    (128, 256, 3, 2, 40), (128, 128, 3, 1, 40), (256, 128, 1, 1, 40), (128, 128, 1, 1, 40), (256, 256, 1, 1, 40), (512, 128, 1, 1, 40),
    (256, 512, 3, 2, 20), (256, 256, 3, 1, 20), (512, 256, 1, 1, 20), (512, 512, 1, 1, 20), (1024, 512, 1, 1, 20),
]
dev = torch.device("cuda:0")
B = int(os.environ.get("WG_BATCH", "8"))
SCALE = int(os.environ.get("WG_SCALE", "1"))          # 2: the YOLOPoint-l channel counts
SHAPES = [(a * SCALE, b * SCALE, k, s_, h) for a, b, k, s_, h in SHAPES]


def view(t, C_):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], 0, C_, 0
    return v


tot = 0.0
for Cin, Cout, k, st, Ho in SHAPES:
    x = torch.randn(B, Ho * st, Ho * st, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(Cin, k, k, Cout, device=dev)
    sp = _hip.stream_ptr()
    vx, vy = view(x, Cin), view(dy, Cout)
    for _ in range(3):
        check(lib().yp_conv_wgrad(vx, vy, _hip.YP_BF16, B, k, st, dw.data_ptr(), sp))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib().yp_conv_wgrad(vx, vy, _hip.YP_BF16, B, k, st, dw.data_ptr(), sp))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * B * Ho * Ho * Cin * k * k * Cout
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    tot += us
    print(f"  {Cin:4d}->{Cout:4d} k{k} s{st} {Ho:3d}^2  {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s  {mb / us * 1e3:6.0f} GB/s(min)")
print(f"  total {tot:.0f} us")
