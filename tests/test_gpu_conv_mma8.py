"""The 8-wave 32x32x16 convolution kernels (csrc/conv_mma8.hip, tile ids 41-44: ping-pong schedule on four tile shapes; 57: two-phase
schedule, the one with the 8-bit instantiation; 58: free-running schedule; 61 / 62: one wavefront per SIMD, 256 / 224 x 256 tiles) on shapes that exercise their corner cases: ragged M and
N tails, one / several filter taps, K of one, two and many 64-deep k tiles (ring prologue / tail), two channel-concatenated sources (one
read through a 2x upsample), residual add, split destination, stride 2, fp32 output, the zero-stuffed input of a stride-2 dgrad and the
BatchNorm-statistics epilogue.  Reference: torch conv2d on the CPU in fp32 on the SAME 16-bit-rounded operands (what the kernel
multiplies): fp32 outputs agree to 2e-5 of max|ref| (accumulation order), 16-bit outputs to one output rounding (2e-3 f16, 1.6e-2 bf16).
Replaces: models/common.py:22-34 (Conv), :79-89 (Bottleneck.cv2) for channel counts that are multiples of 64."""
import pytest
import torch
import torch.nn.functional as F

from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

pytestmark = pytest.mark.gpu
TILES = (41, 42, 43, 44, 57, 58, 61, 62)
STAT_ROW_PX = {41: 128, 57: 128, 58: 128, 61: 256, 62: 224}        # pixels per BatchNorm-statistics row (others: 64)

CASES = {
    "pointwise_256_256_ragged_m": dict(cin=256, cout=256, k=1, s=1, H=21, B=3),
    "pointwise_k64_single_tile": dict(cin=64, cout=128, k=1, s=1, H=20, B=2),
    "pointwise_k128_two_tiles_f32out": dict(cin=128, cout=192, k=1, s=1, H=12, B=2, out_f32=True, act=False),
    "conv3x3_128_residual": dict(cin=128, cout=128, k=3, s=1, H=20, B=2, res=True),
    "conv3x3_stride2_256_512": dict(cin=256, cout=512, k=3, s=2, H=10, B=2),
    "concat_upsample_512": dict(cin=(256, 256), ups0=True, cout=256, k=1, s=1, H=16, B=2),
    "concat_3x3_two_sources": dict(cin=(64, 128), cout=64, k=3, s=1, H=11, B=2),
    "split_destination": dict(cin=128, cout=256, k=1, s=1, H=24, B=2, split=128),
    "ragged_n_72": dict(cin=64, cout=72, k=3, s=1, H=9, B=1),
}


def _reference(x_list, ups0, w, b, k, s, act, res, dt):
    xs = []
    for i, x in enumerate(x_list):
        t = x.float().permute(0, 3, 1, 2)
        if i == 0 and ups0:
            t = F.interpolate(t, scale_factor=2, mode="nearest")
        xs.append(t)
    y = F.conv2d(torch.cat(xs, 1), w.to(dt).float(), b, s, k // 2)
    if act:
        y = F.silu(y)
    if res is not None:
        y = y + res.float().permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_every_tile(cuda, name, dtype):
    c = CASES[name]
    code = _hip.dtype_code(dtype)
    dt = _hip.torch_dtype(code)
    cins = c["cin"] if isinstance(c["cin"], tuple) else (c["cin"],)
    cout, k, s, Ho, B = c["cout"], c["k"], c["s"], c["H"], c["B"]
    Hi = Ho * s
    out_f32, act, ups0, split = c.get("out_f32", False), c.get("act", True), c.get("ups0", False), c.get("split")
    g = torch.Generator().manual_seed(len(name))
    w = torch.randn(cout, sum(cins), k, k, generator=g) * (1.5 / (sum(cins) * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    xs_cpu = [torch.randn(B, Hi >> (1 if (i == 0 and ups0) else 0), Hi >> (1 if (i == 0 and ups0) else 0), ci, generator=g).to(dt) for i, ci in enumerate(cins)]
    res_cpu = torch.randn(B, Ho, Ho, cout, generator=g).to(dt) if c.get("res") else None
    ref = _reference(xs_cpu, ups0, w, b, k, s, act, res_cpu, dt)
    scale = float(ref.abs().max())
    for tile in TILES:
        pb = PlanBuilder(B, code, cuda)
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
            kw = dict(out=pb.new_buf(Ho, Ho, split).view(), out2=pb.new_buf(Ho, Ho, cout - split).view())
        out = pb.conv(views, w, b, k, s, k // 2, _hip.YP_ACT_SILU if act else _hip.YP_ACT_NONE, res=resv, out_f32=out_f32, tile=tile, **kw)
        plan = pb.finish()
        plan.run()
        plan.run()                                         # (a second replay: the ring / barrier state of the first must not leak)
        torch.cuda.synchronize()
        if split:
            got = torch.cat((kw["out"].buf.t[..., :split].float().cpu(), kw["out2"].buf.t[..., :cout - split].float().cpu()), -1)
        else:
            got = out.buf.t[..., :cout].float().cpu()
        err = float((got - ref).abs().max()) / scale
        bar = 2e-5 if out_f32 else (2e-3 if dtype == "f16" else 1.6e-2)
        assert err < bar, (name, tile, err)


@pytest.mark.parametrize("tile", TILES)
def test_matches_the_first_generation_kernel_on_a_deep_layer(cuda, tile):
    """3x3 256 -> 256 at 40x40, batch 4 (K = 2304: 36 k tiles, 9 taps x 4 tiles): against the 4-wave generic kernel (tile 3) on the same
    buffers -- both accumulate in fp32 over the same 16-bit operands, only the summation order differs."""
    B, C, Ho = 4, 256, 40
    g = torch.Generator().manual_seed(5)
    w = torch.randn(C, C, 3, 3, generator=g) * (1.5 / (C * 9) ** 0.5)
    b = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, Ho, Ho, C, generator=g).half()
    outs = []
    for tl in (3, tile):
        pb = PlanBuilder(B, _hip.YP_F16, cuda)
        pb.autotune = False
        buf = pb.new_buf(Ho, Ho, C)
        buf.t.copy_(x.to(cuda))
        out = pb.conv(buf.view(), w, b, 3, 1, 1, _hip.YP_ACT_NONE, out_f32=True, tile=tl)
        plan = pb.finish()
        plan.run()
        torch.cuda.synchronize()
        outs.append(out.buf.t[..., :C].float().cpu())
    err = float((outs[0] - outs[1]).abs().max()) / float(outs[0].abs().max())
    assert err < 2e-5, (tile, err)


@pytest.mark.parametrize("tile", TILES)
def test_zero_stuffed_input_of_a_stride2_dgrad(cuda, tile):
    """dgrad of a 3x3 / stride 2 convolution = stride-1 convolution of the zero-stuffed output gradient with the flipped filter: the
    kernel reads dy through `in0_zero_stuffed` (only even logical rows / columns carry data)."""
    B, C, Hs = 2, 128, 12
    g = torch.Generator().manual_seed(9)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    dy = torch.randn(B, Hs, Hs, C, generator=g).half()
    stuffed = torch.zeros(B, 2 * Hs, 2 * Hs, C)
    stuffed[:, ::2, ::2] = dy.float()
    ref = F.conv2d(stuffed.permute(0, 3, 1, 2), w.half().float(), None, 1, 1).permute(0, 2, 3, 1)
    pb = PlanBuilder(B, _hip.YP_F16, cuda)
    pb.autotune = False
    buf = pb.new_buf(Hs, Hs, C)
    buf.t.copy_(dy.to(cuda))
    out = pb.conv(buf.view(), w, None, 3, 1, 1, _hip.YP_ACT_NONE, out_f32=True, tile=tile, extra=dict(zero_stuffed=True))
    plan = pb.finish()
    plan.run()
    torch.cuda.synchronize()
    got = out.buf.t[..., :C].float().cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, (tile, err)


@pytest.mark.parametrize("tile", TILES)
def test_batchnorm_statistics_epilogue(cuda, tile):
    """Training forward: the raw output plus per-row-block column sums / sums of squares (`bn_partial`, laid out [2][C][rows]); folded they are the batch
    statistics of the stored tensor's fp32 source.  Deterministic: two runs give bit-identical partial rows."""
    B, Cin, Cout, Ho = 3, 128, 192, 14          # M = 588: ragged against every row-block size
    g = torch.Generator().manual_seed(11)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (Cin * 9) ** 0.5)
    x = torch.randn(B, Ho, Ho, Cin, generator=g).to(torch.bfloat16)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    runs = []
    for _ in range(2):
        pb = PlanBuilder(B, _hip.YP_BF16, cuda)
        pb.autotune = False
        buf = pb.new_buf(Ho, Ho, Cin)
        buf.t.copy_(x.to(cuda))
        part = torch.zeros(((B * Ho * Ho + 63) // 64, 2, Cout), dtype=torch.float32, device=cuda)
        out = pb.conv(buf.view(), w, None, 3, 1, 1, _hip.YP_ACT_NONE, tile=tile, extra=dict(bn_partial=part))
        rows = pb.last_bn_rows
        plan = pb.finish()
        plan.run()
        torch.cuda.synchronize()
        assert rows == (B * Ho * Ho + STAT_ROW_PX.get(tile, 64) - 1) // STAT_ROW_PX.get(tile, 64)
        flat = part.flatten()                              # the kernel's layout: [2][Cout][rows] at the front of the buffer
        assert flat.numel() == 2 * Cout * rows or float(flat[2 * Cout * rows:].abs().max()) == 0.0
        runs.append((flat[:2 * Cout * rows].view(2, Cout, rows).cpu().clone(), out.buf.t[..., :Cout].float().cpu().reshape(-1, Cout)))
    (p0, y0), (p1, _) = runs
    assert torch.equal(p0, p1)
    assert float((y0 - ref).abs().max()) / float(ref.abs().max()) < 1.6e-2
    s1, s2 = p0[0].sum(1), p0[1].sum(1)
    assert float((s1 - ref.sum(0)).abs().max()) / float(ref.sum(0).abs().max()) < 1e-4
    assert float((s2 - (ref * ref).sum(0)).abs().max()) / float((ref * ref).sum(0).abs().max()) < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("Cin,Cout,B,Ho,Wo,tile", [(32, 64, 2, 10, 14, 0), (64, 128, 3, 8, 8, 4), (128, 128, 2, 12, 10, 44), (128, 256, 1, 9, 7, 3)])
def test_stride2_dgrad_as_four_parity_class_convolutions(cuda, dtype, Cin, Cout, B, Ho, Wo, tile):
    """Gradient of Conv(Cin, Cout, 3, 2, 1) w.r.t. its input (reference: autograd through models/common.py:22-34 at stride 2): the input pixels
    of parity (py, px) receive (1 + py) x (1 + px) of the nine taps, so the gradient is four small stride-1 convolutions over dy
    (yp_pack_weight modes 4..7) whose epilogues write their parity class of the [2 Ho][2 Wo] tensor (YpConvDesc.out_phase) -- against
    torch's conv_transpose2d on the same rounded operands, overwriting and accumulating into an existing gradient."""
    from yolopoint_amd.plan import MasterWeight, View
    if dtype == "f32" and tile == 44:
        pytest.skip("the 8-wave kernel has no fp32 instantiation")
    code = _hip.dtype_code(dtype)
    dt = _hip.torch_dtype(code)
    g = torch.Generator().manual_seed(Cin + Ho)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (Cout * 9) ** 0.5)).to(cuda)
    dy = torch.randn(B, Ho, Wo, Cout, generator=g).to(dt)
    prev = torch.randn(B, 2 * Ho, 2 * Wo, Cin, generator=g).to(dt)
    wq = w.cpu().to(dt).float() if dtype != "f32" else w.cpu()
    ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wq, None, 2, 1, output_padding=1).permute(0, 2, 3, 1)
    for acc in (False, True):
        pb = PlanBuilder(B, code, cuda)
        pb.autotune = False
        src = pb.new_buf(Ho, Wo, Cout)
        src.t.copy_(dy.to(cuda))
        dst = pb.new_buf(2 * Ho, 2 * Wo, Cin)
        dst.t.copy_(prev.to(cuda))
        for py in (0, 1):
            for px in (0, 1):
                ph = View(dst, 0, Cin, 0, geom=(Ho, Wo, dst.C))
                pb.conv([src.view()], MasterWeight(w, mode=("phase", py, px), c0=0, cj=Cin, cout_pad=Cout), None, 0, 1, 0, _hip.YP_ACT_NONE, out=ph,
                        res=ph if acc else None, tile=tile, extra=dict(out_phase=(py, px), out_hw=(Ho, Wo)))
        plan = pb.finish()
        plan.run()
        torch.cuda.synchronize()
        got = dst.t.float().cpu()
        want = ref + (prev.float() if acc else 0.0)
        err = float((got - want).abs().max()) / float(want.abs().max())
        assert err < (2e-5 if dtype == "f32" else 1.6e-2), (acc, err)
