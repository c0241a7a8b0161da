"""GPU parity of the conv / block kernels against the CPU oracle (called through the C ABI via
the module API).  Tolerances: TOL in helpers.py (fp32 path 1e-4, f16 4e-3, bf16 3e-2 of max|ref|)."""
import pytest
import torch

from helpers import block_state, rel_err, TOL
from oracle import net_oracle
from yolopoint_amd import _hip
from yolopoint_amd.models.common import Conv, Bottleneck, C3, SPPF
from yolopoint_amd.plan import PlanBuilder, pack_input, unpack_nchw, round_up

pytestmark = pytest.mark.gpu

CONV_CASES = [
    # c1, c2, k, s, p, B, H, W
    (3, 16, 6, 2, 2, 2, 64, 64),      # stem (paired-pixel path for 16-bit types)
    (3, 32, 6, 2, 2, 1, 96, 160),
    (16, 32, 3, 2, 1, 2, 32, 32),
    (32, 32, 1, 1, 0, 2, 40, 24),
    (64, 64, 3, 1, 1, 1, 20, 20),
    (128, 64, 1, 1, 0, 3, 17, 13),    # ragged M
    (64, 128, 3, 2, 1, 2, 18, 22),
    (256, 512, 3, 2, 1, 1, 12, 12),
    (512, 256, 1, 1, 0, 1, 7, 9),
    (8, 8, 3, 1, 1, 1, 9, 9),         # C3 hidden width of YOLOPoint-n
]


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_block(cuda, case, dtype):
    c1, c2, k, s, p, B, H, W = case
    m = Conv(c1, c2, k, s, p).eval()
    m.compute_dtype = dtype
    sd = block_state(m, 11, "blk.")
    x = net_oracle.synth_image(B, c1, H, W, 5) - 0.5
    ref = net_oracle.conv_block(sd, "blk", x, k, s, p)
    got = m.to(cuda)(x.to(cuda))
    e_max, e_l2 = rel_err(got, ref)
    assert e_max < TOL[dtype], (case, dtype, e_max, e_l2)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_conv_every_tile_config(cuda, tile, dtype):
    """Force each tile template on a shape with ragged M / N tails, a residual and a sliced output."""
    torch.manual_seed(0)
    B, H, W, c1, c2 = 2, 19, 23, 64, 136
    code = _hip.dtype_code(dtype)
    w = torch.randn(c2, c1, 3, 3) * 0.05
    b = torch.randn(c2) * 0.1
    x = torch.randn(B, c1, H, W)
    r = torch.randn(B, c2, H, W)
    ref = torch.nn.functional.silu(torch.nn.functional.conv2d(x, w, b, 1, 1)) + r
    pb = PlanBuilder(B, code, cuda)
    xin = pb.new_buf(H, W, c1)
    res = pb.new_buf(H, W, c2)
    big = pb.new_buf(H, W, c2 + 16)
    out = pb.conv(xin.view(), w, b, 3, 1, 1, _hip.YP_ACT_SILU, out=big.view(8, c2), res=res.view(), tile=tile)
    plan = pb.finish()
    pack_input(x.to(cuda), xin.view(), code)
    pack_input(r.to(cuda), res.view(), code)
    plan.run()
    got = unpack_nchw(out, code, B, c2)
    e_max, _ = rel_err(got, ref)
    assert e_max < TOL[dtype], (tile, dtype, e_max)
    # the 8 channels either side of the slice must stay untouched (zero)
    assert float(big.t[..., :8].abs().max()) == 0.0 and float(big.t[..., 8 + c2:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_conv_two_sources_and_upsample(cuda, dtype):
    """cat((ups(a), b), 1) -> 1x1 conv without materialising either the upsample or the concat."""
    torch.manual_seed(1)
    B, H, W = 2, 10, 14
    code = _hip.dtype_code(dtype)
    a = torch.randn(B, 32, H // 2, W // 2)
    b2 = torch.randn(B, 24, H, W)
    w = torch.randn(40, 56, 3, 3) * 0.05
    ref = torch.nn.functional.conv2d(torch.cat((torch.nn.functional.interpolate(a, scale_factor=2, mode="nearest"), b2), 1), w, None, 1, 1)
    pb = PlanBuilder(B, code, cuda)
    ba, bb = pb.new_buf(H // 2, W // 2, 32), pb.new_buf(H, W, 24)
    out = pb.conv([ba.view().up(), bb.view()], w, None, 3, 1, 1, _hip.YP_ACT_NONE)
    plan = pb.finish()
    pack_input(a.to(cuda), ba.view(), code)
    pack_input(b2.to(cuda), bb.view(), code)
    plan.run()
    e_max, _ = rel_err(unpack_nchw(out, code, B, 40), ref)
    assert e_max < TOL[dtype], e_max


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("kind", ["bottleneck", "c3_n1", "c3_n3", "sppf"])
def test_blocks(cuda, kind, dtype):
    B, H, W = 2, 20, 20
    if kind == "bottleneck":
        m, c1 = Bottleneck(32, 32, True, e=1.0), 32
        fn = lambda sd, x: net_oracle.bottleneck(sd, "blk", x)
    elif kind == "c3_n1":
        m, c1 = C3(64, 64, 1), 64
        fn = lambda sd, x: net_oracle.c3(sd, "blk", x, 1)
    elif kind == "c3_n3":
        m, c1 = C3(128, 64, 3), 128
        fn = lambda sd, x: net_oracle.c3(sd, "blk", x, 3)
    else:
        m, c1 = SPPF(128, 128, 5), 128
        fn = lambda sd, x: net_oracle.sppf(sd, "blk", x)
    m.eval()
    for mod in m.modules():
        if hasattr(mod, "compute_dtype"):
            mod.compute_dtype = dtype
    sd = block_state(m, 3, "blk.")
    x = net_oracle.synth_image(B, c1, H, W, 9) - 0.5
    ref = fn(sd, x)
    got = m.to(cuda)(x.to(cuda))
    e_max, _ = rel_err(got, ref)
    assert e_max < TOL[dtype] * 2, (kind, dtype, e_max)


HALO_CASES = [
    # c1, c2, stride, B, H, W, tile(10: BN=32, 11: BN=64, 12: BN=128 on 8x16 pixel tiles, 13..15 the same on 4x16, 0: auto)
    (32, 32, 1, 2, 24, 40, 10), (64, 64, 1, 1, 20, 20, 11), (64, 64, 1, 2, 19, 23, 0), (128, 128, 1, 1, 17, 33, 12),
    (32, 64, 2, 2, 32, 48, 11), (64, 128, 2, 1, 22, 18, 12), (128, 64, 2, 1, 40, 40, 0), (256, 256, 1, 1, 9, 11, 0),
    (64, 72, 1, 1, 16, 16, 12),       # Cout not a multiple of the channel tile
    # 4 x 16 pixel tiles (ids 13..15), incl. single-chunk inputs (one halo buffer in LDS)
    (32, 32, 1, 2, 22, 40, 13), (32, 64, 2, 2, 32, 48, 14), (64, 128, 2, 1, 22, 18, 15), (128, 128, 1, 1, 17, 33, 15), (256, 64, 2, 1, 18, 22, 14),
    (32, 64, 2, 1, 40, 24, 11),       # single-chunk input on the 8 x 16 tile
]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3x3_halo_kernel(cuda, case, dtype):
    """The LDS halo-reuse 3x3 kernel against torch fp32 conv: ragged tiles, both strides, every channel tile,
    residual + sliced output, and equality with the generic kernel's result within 16-bit rounding."""
    c1, c2, s, B, H, W, tile = case
    torch.manual_seed(c1 + c2 + H)
    code = _hip.dtype_code(dtype)
    w = torch.randn(c2, c1, 3, 3) * (1.0 / (3 * c1 ** 0.5))
    b = torch.randn(c2) * 0.1
    x = torch.randn(B, c1, H, W)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    r = torch.randn(B, c2, Ho, Wo)
    ref = torch.nn.functional.silu(torch.nn.functional.conv2d(x, w, b, s, 1)) + r
    outs = {}
    for name, t in (("halo", tile), ("generic", 4)):
        pb = PlanBuilder(B, code, cuda)
        xin, res = pb.new_buf(H, W, c1), pb.new_buf(Ho, Wo, round_up(c2, 8))
        big = pb.new_buf(Ho, Wo, round_up(c2, 8) + 16)
        out = pb.conv(xin.view(), w, b, 3, s, 1, _hip.YP_ACT_SILU, out=big.view(8, round_up(c2, 8)), res=res.view(), tile=t)
        plan = pb.finish()
        pack_input(x.to(cuda), xin.view(), code)
        pack_input(r.to(cuda), res.view(), code)
        plan.run()
        outs[name] = unpack_nchw(out, code, B, c2)
        assert float(big.t[..., :8].abs().max()) == 0.0 and float(big.t[..., 8 + round_up(c2, 8):].abs().max()) == 0.0
    e_max, _ = rel_err(outs["halo"], ref)
    assert e_max < TOL[dtype], (case, dtype, e_max)
    assert rel_err(outs["halo"], outs["generic"])[0] < TOL[dtype]


BNECK_CASES = [
    # C, B, H, W, tile (0 auto, 10/11/12 = 32/64/128 output channels per workgroup, four wavefronts), shortcut
    (32, 2, 24, 40, 0, True), (32, 1, 19, 23, 10, False), (64, 2, 20, 20, 11, True), (64, 1, 17, 33, 10, True),
    (128, 1, 16, 16, 12, True), (128, 2, 9, 21, 11, False), (128, 1, 40, 40, 10, True),
    # 17 / 18 / 19: the same kernel with EIGHT wavefronts per workgroup (two per SIMD), BN = 32 / 64 / 128
    (32, 2, 24, 40, 17, True), (64, 1, 17, 33, 17, True), (64, 2, 20, 20, 18, False), (128, 1, 40, 40, 17, True), (128, 2, 9, 21, 18, True),
    (128, 1, 16, 16, 19, False), (128, 2, 40, 40, 18, True),
]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("case", BNECK_CASES)
def test_fused_bottleneck_kernel(cuda, case, dtype):
    """Bottleneck with cv1 as the LDS prologue of cv2's 3x3 kernel (hidden tensor never in HBM) against torch fp32 and
    against the two-launch form, which it must reproduce EXACTLY (same 16-bit rounding of the hidden tensor, same
    accumulation order); ragged tiles (image borders inside the halo), every channel tile, sliced output."""
    C_, B, H, W, tile, shortcut = case
    torch.manual_seed(C_ + H + W)
    code = _hip.dtype_code(dtype)
    w1 = torch.randn(C_, C_, 1, 1) * (1.0 / C_ ** 0.5)
    b1 = torch.randn(C_) * 0.2
    w2 = torch.randn(C_, C_, 3, 3) * (1.0 / (3 * C_ ** 0.5))
    b2 = torch.randn(C_) * 0.1
    x = torch.randn(B, C_, H, W)
    F = torch.nn.functional
    ref = F.silu(F.conv2d(F.silu(F.conv2d(x, w1, b1)), w2, b2, 1, 1)) + (x if shortcut else 0)
    outs = {}
    for name in ("fused", "two"):
        pb = PlanBuilder(B, code, cuda)
        pb.autotune = False
        xin = pb.new_buf(H, W, C_ + 8)
        big = pb.new_buf(H, W, C_ + 16)
        xv = xin.view(8, C_)
        res = xv if shortcut else None
        if name == "fused":
            out = pb.conv(xv, w2, b2, 3, 1, 1, _hip.YP_ACT_SILU, out=big.view(8, C_), res=res, tile=tile,
                          extra={"pre": (w1, b1, _hip.YP_ACT_SILU)})
        else:
            t = pb.conv(xv, w1, b1, 1, 1, 0, _hip.YP_ACT_SILU)
            out = pb.conv(t, w2, b2, 3, 1, 1, _hip.YP_ACT_SILU, out=big.view(8, C_), res=res)
        plan = pb.finish()
        assert len(plan.records) == (1 if name == "fused" else 2)
        pack_input(x.to(cuda), xv, code)
        plan.run()
        outs[name] = unpack_nchw(out, code, B, C_)
        assert float(big.t[..., :8].abs().max()) == 0.0 and float(big.t[..., 8 + C_:].abs().max()) == 0.0
    assert rel_err(outs["fused"], ref)[0] < TOL[dtype] * 2, (case, dtype)
    assert torch.equal(outs["fused"], outs["two"]), (case, dtype, float((outs["fused"] - outs["two"]).abs().max()))


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("c1,c2,n,B,H,W", [(64, 64, 1, 2, 24, 40), (128, 128, 2, 1, 19, 23), (256, 128, 1, 2, 16, 16), (64, 64, 1, 1, 9, 21)])
def test_c3_tail_fused_into_last_bottleneck(cuda, c1, c2, n, B, H, W, dtype):
    """C3 with cv3 running inside the last Bottleneck's kernel (hidden widths 32 / 64) against the oracle and against the
    unfused launch sequence, which it must reproduce exactly (same 16-bit rounding points, same k order); ragged tiles."""
    m = C3(c1, c2, n).eval()
    for mod in m.modules():
        if hasattr(mod, "compute_dtype"):
            mod.compute_dtype = dtype
    sd = block_state(m, 5, "blk.")
    x = net_oracle.synth_image(B, c1, H, W, 2) - 0.5
    ref = net_oracle.c3(sd, "blk", x, n)
    outs = {}
    for fused in (True, False):
        C3.fuse_tail = fused
        try:
            m._plans = {} if hasattr(m, "_plans") else None
            m.__dict__.pop("_plan_cache", None)
            mm = C3(c1, c2, n).eval()
            mm.load_state_dict(m.state_dict())
            for mod in mm.modules():
                if hasattr(mod, "compute_dtype"):
                    mod.compute_dtype = dtype
            outs[fused] = mm.to(cuda)(x.to(cuda)).float().cpu()
        finally:
            C3.fuse_tail = True
    assert rel_err(outs[True], ref)[0] < TOL[dtype] * 2
    assert torch.equal(outs[True], outs[False]), float((outs[True] - outs[False]).abs().max())


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,W", [(4, 104, 200), (3, 99, 190), (1, 24, 40)])
def test_persistent_bottleneck_c32_equals_the_one_tile_kernel(cuda, monkeypatch, B, H, W, dtype):
    """Tile 16 (bneck32_persist_kernel: the three filters resident in LDS, a workgroup walks its tiles, the next tile's halo and cv2-branch
    tile prefetched into the other half of a double buffer, shortcut read from the resident halo) against tile 10 (one tile per workgroup):
    the same fragments in the same MFMA order -- bit-identical; sizes with more tiles than resident workgroups (676 / 507 tiles for 512:
    the loop, the buffer flip, ragged right / bottom tiles) and fewer (one tile per workgroup, no prefetch); and against the oracle."""
    from yolopoint_amd import plan as yplan
    c1 = c2 = 64
    m = C3(c1, c2, 1).eval()
    sd = block_state(m, 9, "blk.")
    x = net_oracle.synth_image(B, c1, H, W, 4) - 0.5
    ref = net_oracle.c3(sd, "blk", x, 1)
    outs = {}
    saved = dict(yplan._TUNE_CACHE)
    try:
        for tile in (10, 16):
            monkeypatch.setattr(yplan, "_TUNE_CANDIDATES", (1, 2, 4, 5, tile))      # (the fused launch has exactly one applicable candidate)
            yplan._TUNE_CACHE.clear()
            mm = C3(c1, c2, 1).eval()
            mm.load_state_dict(m.state_dict())
            for mod in mm.modules():
                if hasattr(mod, "compute_dtype"):
                    mod.compute_dtype = dtype
            outs[tile] = mm.to(cuda)(x.to(cuda)).float().cpu()
            fused = [k for k, v in yplan._TUNE_CACHE.items() if v[0] == tile]
            assert fused, (tile, list(yplan._TUNE_CACHE.values()))                  # the fused Bottleneck + tail really ran on this tile id
    finally:
        yplan._TUNE_CACHE.clear()
        yplan._TUNE_CACHE.update(saved)
    assert rel_err(outs[16], ref)[0] < TOL[dtype] * 2
    assert torch.equal(outs[16], outs[10]), float((outs[16] - outs[10]).abs().max())


def test_fused_bottleneck_rejects_unsupported(cuda):
    pb = PlanBuilder(1, _hip.YP_F16, cuda)
    pb.autotune = False
    xin = pb.new_buf(8, 8, 256)
    w1, w2 = torch.zeros(256, 256, 1, 1), torch.zeros(256, 256, 3, 3)
    with pytest.raises(_hip.YpError):
        pb.conv(xin.view(), w2, None, 3, 1, 1, _hip.YP_ACT_SILU, extra={"pre": (w1, None, _hip.YP_ACT_SILU)})
        pb.finish().run()


def test_conv3x3_halo_fp32_head_output(cuda):
    """ConvDesc-style use: 3x3, no bias, no activation, fp32 output from f16 operands."""
    torch.manual_seed(3)
    B, C, H, W = 1, 128, 20, 28
    w = torch.randn(C, C, 3, 3) * 0.03
    x = torch.randn(B, C, H, W)
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    pb = PlanBuilder(B, _hip.YP_F16, cuda)
    xin = pb.new_buf(H, W, C)
    out = pb.conv(xin.view(), w, None, 3, 1, 1, _hip.YP_ACT_NONE, out_f32=True)
    plan = pb.finish()
    pack_input(x.to(cuda), xin.view(), _hip.YP_F16)
    plan.run()
    assert out.buf.t.dtype == torch.float32
    assert rel_err(unpack_nchw(out, _hip.YP_F32, B, C), ref)[0] < TOL["f16"]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("cout,B,H,W", [(16, 2, 64, 64), (32, 1, 96, 160), (48, 1, 32, 64), (64, 2, 48, 80)])
def test_fused_stem_kernel(cuda, cout, B, H, W, dtype):
    """yp_stem_conv (NCHW fp32 image -> Conv 6x6/s2/p2 + bias + SiLU -> NHWC) against the oracle's Conv block."""
    m = Conv(3, cout, 6, 2, 2).eval()
    sd = block_state(m, 17, "blk.")
    x = net_oracle.synth_image(B, 3, H, W, 3) - 0.5
    ref = net_oracle.conv_block(sd, "blk", x, 6, 2, 2)
    code = _hip.dtype_code(dtype)
    pb = PlanBuilder(B, code, cuda)
    w, b = m.folded()
    out, launch = pb.stem(w, b, _hip.YP_ACT_SILU, H, W)
    launch(x.to(cuda).contiguous())
    got = unpack_nchw(out, code, B, cout)
    assert rel_err(got, ref)[0] < TOL[dtype], (cout, dtype, rel_err(got, ref))
