"""GPU parity of the direct NHWC weight-gradient kernel (yp_conv_wgrad: LDS transpose reads, no pixel-major copies)
against torch autograd's conv2d weight gradient on the same 16-bit-rounded operands (fp32 reference)."""
import ctypes as C

import pytest
import torch

from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, lib, check

pytestmark = pytest.mark.gpu

CASES = [
    # Cin, Cout, k, B, H, W, ups, (x channel offset, x cstride extra)
    (32, 32, 1, 2, 24, 40, 0), (64, 128, 1, 1, 17, 13, 0), (128, 64, 1, 3, 20, 20, 1), (256, 256, 1, 1, 10, 10, 0),
    (72, 40, 1, 2, 9, 11, 0),                    # channel counts that are not multiples of the 16 / 64 blocks
    # 1x1 with >= 128 channels on both sides and enough pixels per workgroup: the 128 x 128-block kernel (ragged channel blocks, a source
    # read through its 2x upsample, the many-block / small-split corner)
    (1024, 1024, 1, 2, 40, 40, 0), (200, 136, 1, 2, 160, 160, 0), (256, 128, 1, 4, 160, 160, 1),
    (32, 32, 3, 2, 24, 40, 0), (64, 64, 3, 1, 20, 20, 0), (64, 128, 3, 2, 19, 23, 0), (128, 128, 3, 1, 16, 16, 0),
    (136, 72, 3, 1, 9, 21, 0), (256, 256, 3, 1, 8, 8, 0), (64, 64, 3, 1, 16, 16, 1),
    # 3x3 stride 2 (H, W are the INPUT size; output = ceil / 2): trailing 8th field = stride
    (32, 64, 3, 2, 32, 48, 0, 2), (64, 128, 3, 1, 22, 18, 0, 2), (128, 256, 3, 1, 40, 40, 0, 2), (72, 40, 3, 2, 10, 34, 0, 2),
]


def view(t, coff, C_, ups=0):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], coff, C_, ups
    return v


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", CASES)
def test_wgrad_matches_autograd(cuda, case, dtype):
    Cin, Cout, k, B, H, W, ups = case[:7]
    st = case[7] if len(case) > 7 else 1
    torch.manual_seed(Cin + Cout + H)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = _hip.dtype_code(dtype)
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    if ups:
        H, W = hs * 2, ws * 2
    xbuf = torch.randn(B, hs, ws, Cin + 16, device=cuda).to(td)          # view = channels [8, 8+Cin) of a wider buffer
    dybuf = (torch.randn(B, H // st, W // st, Cout + 8, device=cuda) * 0.1).to(td)   # view = channels [0, Cout)
    dw = torch.zeros(Cin, k, k, Cout, device=cuda)
    check(lib().yp_conv_wgrad(view(xbuf, 8, Cin, ups), view(dybuf, 0, Cout), code, B, k, st, dw.data_ptr(), _hip.stream_ptr()))
    torch.cuda.synchronize()
    x = xbuf[..., 8:8 + Cin].float().permute(0, 3, 1, 2)
    if ups:
        x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    dy = dybuf[..., :Cout].float().permute(0, 3, 1, 2)
    w = torch.zeros(Cout, Cin, k, k, device=cuda, requires_grad=True)
    torch.nn.functional.conv2d(x, w, None, st, k // 2).backward(dy)
    ref = w.grad.permute(1, 2, 3, 0)                                       # [ci][r][s][co]
    err = float((dw - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, (case, dtype, err)      # same products, fp32 accumulation in a different order


def test_block_size_selection(cuda):
    """yp_wgrad_block: 128 x 128 blocks for 1x1 filters with >= 128 channels on both sides when a workgroup keeps >= 12 pixel tiles."""
    def blk(Cin, Cout, k, B, H, W):
        x = torch.empty(1, H, W, Cin, device=cuda, dtype=torch.bfloat16)
        dy = torch.empty(1, H, W, Cout, device=cuda, dtype=torch.bfloat16)
        return lib().yp_wgrad_block(view(x, 0, Cin), view(dy, 0, Cout), B, k)
    assert blk(1024, 1024, 1, 2, 40, 40) == 128 and blk(200, 136, 1, 2, 160, 160) == 128 and blk(256, 128, 1, 4, 160, 160) == 128
    assert blk(256, 256, 1, 16, 40, 40) == 64          # YOLOPoint-s P4 at 8 samples per GPU: too few tiles per workgroup
    assert blk(64, 256, 1, 32, 160, 160) == 64 and blk(256, 256, 3, 32, 80, 80) == 64


def test_wgrad_rejects_bad_arguments(cuda):
    x = torch.zeros(1, 8, 8, 16, device=cuda, dtype=torch.bfloat16)
    dw = torch.zeros(16, 5, 5, 16, device=cuda)
    assert lib().yp_conv_wgrad(view(x, 0, 16), view(x, 0, 16), _hip.YP_BF16, 1, 5, 1, dw.data_ptr(), None) != 0
    assert b"filters only" in lib().yp_last_error()
    assert lib().yp_conv_wgrad(view(x, 0, 16), view(x, 0, 16), _hip.YP_F32, 1, 1, 1, dw.data_ptr(), None) != 0
    assert lib().yp_conv_wgrad(view(x, 0, 16), view(x, 0, 16), _hip.YP_BF16, 1, 1, 2, dw.data_ptr(), None) != 0


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,W,Cout", [(2, 64, 64, 32), (3, 48, 80, 16), (1, 36, 44, 48), (2, 128, 96, 64), (1, 64, 192, 80), (16, 160, 160, 32)])
def test_stem_wgrad_matches_autograd(cuda, B, H, W, Cout, dtype):
    """yp_stem_wgrad (6x6 / stride 2 / pad 2 over the packed 4-channel image; reference models/YOLOPoint.py:156) against torch autograd's
    conv2d weight gradient on the same 16-bit operands; sizes with partial 8 x 16 output patches, every dy block count (Cout 16..80), more
    patches than workgroups (16 x 160 x 160); twice: bit-identical."""
    torch.manual_seed(H + W + Cout)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = _hip.dtype_code(dtype)
    img = torch.zeros(B, H, W, 4, device=cuda)
    img[..., :3] = torch.rand(B, H, W, 3, device=cuda)
    img = img.to(td)
    dybuf = (torch.randn(B, H // 2, W // 2, Cout + 8, device=cuda) * 0.1).to(td)          # view = channels [8, 8+Cout)
    nsl = lib().yp_stem_wgrad_slabs(B, H, W)
    assert nsl > 0
    slabs = torch.full((nsl, 144 * Cout), float("nan"), device=cuda)
    outs = []
    for _ in range(2):
        dw = torch.full((4, 6, 6, Cout), float("nan"), device=cuda)
        check(lib().yp_stem_wgrad(view(img, 0, 4), view(dybuf, 8, Cout), code, B, slabs.data_ptr(), dw.data_ptr(), _hip.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    x = img[..., :3].float().permute(0, 3, 1, 2)
    dy = dybuf[..., 8:8 + Cout].float().permute(0, 3, 1, 2)
    w = torch.zeros(Cout, 3, 6, 6, device=cuda, requires_grad=True)
    torch.nn.functional.conv2d(x, w, None, 2, 2).backward(dy)
    ref = w.grad.permute(1, 2, 3, 0)
    err = float((outs[0][:3] - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err
    assert float(outs[0][3].abs().max()) == 0.0           # the zero channel


def test_stem_wgrad_rejects_bad_arguments(cuda):
    img = torch.zeros(1, 16, 16, 4, device=cuda, dtype=torch.bfloat16)
    dy = torch.zeros(1, 8, 8, 16, device=cuda, dtype=torch.bfloat16)
    s, dw = torch.zeros(64, 144 * 16, device=cuda), torch.zeros(4, 6, 6, 16, device=cuda)
    assert lib().yp_stem_wgrad(view(img, 0, 4), view(dy, 0, 16), _hip.YP_F32, 1, s.data_ptr(), dw.data_ptr(), None) != 0
    assert lib().yp_stem_wgrad(view(dy, 0, 16), view(dy, 0, 16), _hip.YP_BF16, 1, s.data_ptr(), dw.data_ptr(), None) != 0
    assert b"packed 4-channel image" in lib().yp_last_error()
    assert lib().yp_stem_wgrad(view(img, 0, 4), view(img, 0, 4), _hip.YP_BF16, 1, s.data_ptr(), dw.data_ptr(), None) != 0


Q8_CASES = [
    # Cin, Cout, k, B, H, W, ups[, stride]: channel counts are multiples of 64 wherever the training graph uses 8-bit operands; offsets are
    # multiples of 16 bytes
    (64, 64, 1, 2, 24, 40, 0), (128, 64, 1, 3, 20, 20, 1), (256, 256, 1, 1, 10, 10, 0), (64, 128, 1, 1, 17, 13, 0),
    (1024, 1024, 1, 2, 40, 40, 0), (256, 128, 1, 4, 160, 160, 1),                          # 128 x 128 blocks
    (64, 64, 3, 1, 20, 20, 0), (64, 128, 3, 2, 19, 23, 0), (128, 128, 3, 1, 16, 16, 0), (256, 256, 3, 1, 8, 8, 0), (64, 64, 3, 1, 16, 16, 1),
    (64, 128, 3, 1, 22, 18, 0, 2), (128, 256, 3, 1, 40, 40, 0, 2), (64, 64, 3, 2, 32, 48, 0, 2),
]


@pytest.mark.parametrize("case", Q8_CASES)
def test_wgrad_q8_matches_autograd_on_the_same_bytes(cuda, case):
    """yp_conv_wgrad_q8 (x = e4m3 bytes, dy = e5m2 bytes, per-tensor scales; ds_read_b64_tr_b8 fragments, v_mfma_f32_16x16x32_fp8_bf8) against
    torch autograd's conv2d weight gradient on the SAME 8-bit operands (torch.float8_e4m3fn / float8_e5m2 widened to fp32, times the scales):
    the kernel's only freedom is the fp32 accumulation order."""
    Cin, Cout, k, B, H, W, ups = case[:7]
    st = case[7] if len(case) > 7 else 1
    torch.manual_seed(Cin + Cout + H + k)
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    if ups:
        H, W = hs * 2, ws * 2
    sx, sdy = 0.37, 2.5e-4
    x8 = (torch.randn(B, hs, ws, Cin + 32, device=cuda) * 2.0).clamp(-440, 440).to(torch.float8_e4m3fn)       # view = channels [16, 16 + Cin)
    dy8 = (torch.randn(B, H // st, W // st, Cout + 16, device=cuda) * 30.0).to(torch.float8_e5m2)              # view = channels [0, Cout)
    scales = torch.tensor([sx, sdy], device=cuda)
    dw = torch.zeros(Cin, k, k, Cout, device=cuda)
    xb, dyb = x8.view(torch.uint8), dy8.view(torch.uint8)
    check(lib().yp_conv_wgrad_q8(view(xb, 16, Cin, ups), view(dyb, 0, Cout), scales.data_ptr(), scales.data_ptr() + 4, B, k, st, dw.data_ptr(), _hip.stream_ptr()))
    torch.cuda.synchronize()
    # fp64 reference (the products of two 8-bit values are exact in fp32; what differs is the order of the fp32 sums)
    x = (x8[..., 16:16 + Cin].double() * sx).permute(0, 3, 1, 2)
    if ups:
        x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    dy = (dy8[..., :Cout].double() * sdy).permute(0, 3, 1, 2)
    w = torch.zeros(Cout, Cin, k, k, device=cuda, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, w, None, st, k // 2).backward(dy)
    ref = w.grad.permute(1, 2, 3, 0)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    # measured 0.3 - 4.0e-5 over these cases (the 16-bit kernel: < 2e-5): v_mfma_f32_16x16x32_fp8_bf8 adds its 32 exact products with less than
    # fp32 precision inside the instruction before the fp32 accumulate -- two orders of magnitude below the 8-bit operands' own resolution
    assert err < 1e-4, (case, err)


def test_wgrad_q8_rejects_unaligned_views(cuda):
    x = torch.zeros(1, 8, 8, 72, device=cuda, dtype=torch.uint8)
    dw = torch.zeros(64, 1, 1, 64, device=cuda)
    s = torch.ones(2, device=cuda)
    assert lib().yp_conv_wgrad_q8(view(x, 8, 64), view(x, 0, 64), s.data_ptr(), s.data_ptr() + 4, 1, 1, 1, dw.data_ptr(), None) != 0
    assert b"16-channel aligned" in lib().yp_last_error()
    assert lib().yp_conv_wgrad_q8(view(x, 0, 64), view(x, 0, 64), None, s.data_ptr(), 1, 1, 1, dw.data_ptr(), None) != 0
