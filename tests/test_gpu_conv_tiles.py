"""Every tile / main-loop variant of the generic implicit-GEMM convolution kernel (ids 1-5: first-generation loop, 21-27: several k
tiles per barrier + register double-buffered fragments, 31 / 33: waves split k; 71-73: the wave-private split-K kernels of
csrc/conv_wsk.hip) on the layer shapes that exercise its corner cases:
ragged M and N tails, N = 65 with fp32 output (the keypoint head), K that is not a multiple of the stage depth, two channel-
concatenated sources (one read through a 2x upsample), residual add, split destinations, stride 2, and the Detect decode epilogue.
The plan-time autotuner picks among these per layer, so each must be right on its own.  Reference: torch conv2d on the CPU in fp32
on the SAME 16-bit-rounded operands (what the kernel multiplies): fp32-output cases agree to 2e-5 of max|ref| (fp32 accumulation
order), 16-bit outputs to one output rounding (2e-3)."""
import os

import pytest
import torch
import torch.nn.functional as F

from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

pytestmark = pytest.mark.gpu
TILES = (1, 2, 3, 4, 5, 21, 22, 23, 24, 25, 26, 27, 31, 33) + ((71, 72, 73, 74, 75, 76) if "libPW" in os.environ.get("YP_HIP_LIB", "") else ())

# name: Cin (or (C0, C1) for two sources), Cout, k, stride, Hout, batch, extras
CASES = {
    "keypoint_head_128_65_f32out": dict(cin=128, cout=65, k=1, s=1, H=40, B=3, out_f32=True, act=False),
    "pointwise_256_256": dict(cin=256, cout=256, k=1, s=1, H=20, B=2),
    "pointwise_k160_ragged": dict(cin=160, cout=40, k=1, s=1, H=9, B=5),
    "pointwise_k96_three_tiles": dict(cin=96, cout=72, k=1, s=1, H=13, B=2),
    "concat_upsample_512": dict(cin=(256, 256), ups0=True, cout=256, k=1, s=1, H=16, B=2),
    "split_destination": dict(cin=128, cout=128, k=1, s=1, H=24, B=2, split=64),
    "conv3x3_residual": dict(cin=64, cout=64, k=3, s=1, H=20, B=2, res=True),
    "conv3x3_stride2_deep": dict(cin=256, cout=512, k=3, s=2, H=10, B=2),
    "conv3x3_k288": dict(cin=32, cout=64, k=3, s=2, H=24, B=1),
    # the wave-private split-K tiles (71-73): fewer k tiles than waves, many tiles per wave across filter taps and sources, ragged tails
    "pointwise_k64_two_tiles": dict(cin=64, cout=96, k=1, s=1, H=11, B=3),
    "conv3x3_deep_256": dict(cin=256, cout=256, k=3, s=1, H=20, B=2),
    "conv3x3_concat_stride2": dict(cin=(64, 96), cout=72, k=3, s=2, H=7, B=3),
    "pointwise_1024_512": dict(cin=1024, cout=512, k=1, s=1, H=10, B=2),
}


def _reference(x_list, ups0, w, b, k, s, act, res):
    xs = []
    for i, x in enumerate(x_list):
        t = x.float().permute(0, 3, 1, 2)
        if i == 0 and ups0:
            t = F.interpolate(t, scale_factor=2, mode="nearest")
        xs.append(t)
    y = F.conv2d(torch.cat(xs, 1), w.half().float(), b, s, k // 2)
    if act:
        y = F.silu(y)
    if res is not None:
        y = y + res.float().permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("name", list(CASES))
def test_every_tile_variant(cuda, name):
    c = CASES[name]
    cins = c["cin"] if isinstance(c["cin"], tuple) else (c["cin"],)
    cout, k, s, Ho, B = c["cout"], c["k"], c["s"], c["H"], c["B"]
    Hi = Ho * s
    out_f32, act, ups0, split = c.get("out_f32", False), c.get("act", True), c.get("ups0", False), c.get("split")
    g = torch.Generator().manual_seed(len(name))
    w = torch.randn(cout, sum(cins), k, k, generator=g) * (1.5 / (sum(cins) * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    xs_cpu = [torch.randn(B, Hi >> (1 if (i == 0 and ups0) else 0), Hi >> (1 if (i == 0 and ups0) else 0), ci, generator=g).half() for i, ci in enumerate(cins)]
    res_cpu = torch.randn(B, Ho, Ho, cout, generator=g).half() if c.get("res") else None
    ref = _reference(xs_cpu, ups0, w, b, k, s, act, res_cpu)
    scale = float(ref.abs().max())
    ran = []
    for tile in TILES:
        pb = PlanBuilder(B, _hip.YP_F16, cuda)
        pb.autotune = False
        views = []
        for i, xc in enumerate(xs_cpu):
            buf = pb.new_buf(xc.shape[1], xc.shape[2], xc.shape[3])
            buf.t.copy_(xc.to(cuda))
            views.append(buf.view().up() if (i == 0 and ups0) else buf.view())
        resv = None
        if res_cpu is not None:
            rb = pb.new_buf(Ho, Ho, cout)
            rb.t.copy_(res_cpu.to(cuda))
            resv = rb.view()
        kw = {}
        if split:
            o1, o2 = pb.new_buf(Ho, Ho, split), pb.new_buf(Ho, Ho, cout)
            kw = dict(out=o1.view(), out2=o2.view(cout - split, split) if False else pb.new_buf(Ho, Ho, cout - split).view())
        try:
            out = pb.conv(views, w, b, k, s, k // 2, _hip.YP_ACT_SILU if act else _hip.YP_ACT_NONE, res=resv, out_f32=out_f32, tile=tile, **kw)
        except _hip.YpError:
            continue                                   # the variant does not apply to this convolution (e.g. 128-row tiles need nothing special; fp32 ...)
        plan = pb.finish()
        plan.run()
        torch.cuda.synchronize()
        if split:
            got = torch.cat((kw["out"].buf.t[..., :split].float().cpu(), kw["out2"].buf.t[..., :cout - split].float().cpu()), -1)
        else:
            got = out.buf.t[..., :cout].float().cpu()
        err = float((got - ref).abs().max()) / scale
        bar = 2e-5 if out_f32 else 2e-3
        assert err < bar, (name, tile, err)
        ran.append(tile)
    assert {1, 4, 24, 31} <= set(ran), ran # both generations, the waves-split-k loop and the wave-private split-K kernels were exercised
