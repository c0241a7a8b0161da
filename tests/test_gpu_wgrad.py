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


def test_wgrad_rejects_bad_arguments(cuda):
    x = torch.zeros(1, 8, 8, 16, device=cuda, dtype=torch.bfloat16)
    dw = torch.zeros(16, 5, 5, 16, device=cuda)
    assert lib().yp_conv_wgrad(view(x, 0, 16), view(x, 0, 16), _hip.YP_BF16, 1, 5, 1, dw.data_ptr(), None) != 0
    assert b"filters only" in lib().yp_last_error()
    assert lib().yp_conv_wgrad(view(x, 0, 16), view(x, 0, 16), _hip.YP_F32, 1, 1, 1, dw.data_ptr(), None) != 0
    assert lib().yp_conv_wgrad(view(x, 0, 16), view(x, 0, 16), _hip.YP_BF16, 1, 1, 2, dw.data_ptr(), None) != 0
