"""Review item 4 (BatchNorm apply out of HBM), measured: a 1x1 convolution that applies x = silu(scale * y + shift) to its input tile in LDS
(probe build -DYP_PROBE_BNFUSE: between the DMA's landing and the fragment reads) against today's two launches -- the apply pass
(yp_bn_act_apply: read y, write x) followed by the plain convolution -- on the 1x1 consumer shapes inside the C3 blocks of YOLOPoint-l at 32 images
(the training forward's pair pass at 16 samples per GPU) and of YOLOPoint-s at 16 images.
YP_HIP_LIB=yolopoint_amd/lib/ab/libBN.so python tools/probe/bnfuse_bench.py [l|s]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, lib, check
from yolopoint_amd.plan import PlanBuilder

which = sys.argv[1] if len(sys.argv) > 1 else "l"
B, w = (32, 2) if which == "l" else (16, 1)
dev = torch.device("cuda:0")
l = lib()
l.yp_debug_set_xform.argtypes = [C.c_void_p, C.c_void_p]
code = _hip.YP_BF16
# (H, Cin, Cout): cv1+cv2 of a C3 (c1 -> 2 c_), m.k.cv1 (c_ -> c_), cv3 (2 c_ -> c2) at the four pyramid levels
shapes = [(160, 64 * w, 64 * w), (160, 32 * w, 32 * w), (80, 128 * w, 128 * w), (80, 64 * w, 64 * w), (40, 256 * w, 256 * w), (40, 128 * w, 128 * w),
          (20, 512 * w, 512 * w), (20, 256 * w, 256 * w)]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def view(t, C_):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], 0, C_, 0
    return v


print(f"# YOLOPoint-{which}, {B} images, bf16; us per launch (HIP events, back to back).  apply = BatchNorm apply + SiLU pass (read y, write x); conv = 1x1 convolution "
      f"reading x (tile = the faster of 128x64 / 64x64, first-generation loop); fused = the same convolution reading y and applying scale/shift/SiLU in its LDS tile")
print(f"{'H':>4} {'Cin':>5} {'Cout':>5} {'MB(y)':>7} | {'apply':>7} {'conv':>7} {'apply+conv':>10} | {'fused':>7} {'fused/conv':>10} {'gain us':>8} | max|fused - two-launch| / max")
tot = [0.0, 0.0, 0.0]
for H, Cin, Cout in shapes:
    torch.manual_seed(H + Cin)
    wgt = torch.randn(Cout, Cin, 1, 1) * (1.0 / Cin ** 0.5)
    best = None
    for tile in (2, 4):
        pb = PlanBuilder(B, code, dev); pb.autotune = False
        y = pb.new_buf(H, H, Cin); y.t.normal_()
        x = pb.new_buf(H, H, Cin)
        o1 = pb.conv(x.view(), wgt, None, 1, 1, 0, _hip.YP_ACT_NONE, tile=tile)
        p_plain = pb.finish()
        pb2 = PlanBuilder(B, code, dev); pb2.autotune = False
        y2 = pb2.new_buf(H, H, Cin); y2.t.copy_(y.t)
        o2 = pb2.conv(y2.view(), wgt, None, 1, 1, 0, _hip.YP_ACT_NONE, tile=tile)
        p_fused = pb2.finish()
        gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2
        mean, invstd = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.7
        scale = (gamma * invstd).contiguous(); shift = (beta - mean * gamma * invstd).contiguous()
        none = YpView()
        apply = lambda: check(l.yp_bn_act_apply(view(y.t, Cin), view(x.t, Cin), none, code, B, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, _hip.stream_ptr()))
        assert l.yp_debug_set_xform(None, None) == 0
        t_apply = timeit(apply)
        t_conv = timeit(p_plain.run)
        apply(); p_plain.run(); torch.cuda.synchronize()
        ref = o1.buf.t[..., :Cout].float().clone()
        assert l.yp_debug_set_xform(scale.data_ptr(), shift.data_ptr()) == 0
        t_fused = timeit(p_fused.run)
        p_fused.run(); torch.cuda.synchronize()
        got = o2.buf.t[..., :Cout].float()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert l.yp_debug_set_xform(None, None) == 0
        if best is None or t_apply + t_conv < best[0] + best[1]:
            best = (t_apply, t_conv, t_fused, err, tile)
        del pb, pb2, p_plain, p_fused
    ta, tc, tf, err, tile = best
    tot[0] += ta; tot[1] += tc; tot[2] += tf
    print(f"{H:4d} {Cin:5d} {Cout:5d} {B * H * H * Cin * 2 / 1e6:7.1f} | {ta:7.1f} {tc:7.1f} {ta + tc:10.1f} | {tf:7.1f} {tf / tc:10.2f} {ta + tc - tf:8.1f} | {err:.1e} (tile {tile})")
print(f"# sum over the shapes: apply {tot[0]:.0f} + conv {tot[1]:.0f} = {tot[0] + tot[1]:.0f} us; fused {tot[2]:.0f} us")
