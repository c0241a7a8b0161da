"""GPU parity of the training BatchNorm kernels (reference models/common.py:18-34: BatchNorm2d(eps=1e-3, momentum=0.03) + SiLU in
train mode, and its autograd backward) against torch on the same 16-bit-rounded tensors."""
import pytest
import torch

from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, lib, check

pytestmark = pytest.mark.gpu


def view(t, coff, C_):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], coff, C_, 0
    return v


NULLV = YpView()


@pytest.mark.parametrize("C,B,H,W,act,with_res", [(32, 8, 80, 80, 1, False), (64, 2, 20, 24, 1, True), (256, 3, 10, 10, 0, False), (8, 1, 33, 7, 1, False),
                                                   (512, 2, 5, 5, 1, True)])
def test_bn_forward_backward(cuda, C, B, H, W, act, with_res):
    torch.manual_seed(C + H)
    l, st = lib(), _hip.stream_ptr()
    code = _hip.YP_BF16
    raw = (torch.randn(B, H, W, C + 8, device=cuda) * 1.7 + 0.3).to(torch.bfloat16)
    res = torch.randn(B, H, W, C, device=cuda).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, C, device=cuda) * 0.1).to(torch.bfloat16)
    gamma, beta = torch.rand(C, device=cuda) + 0.5, torch.randn(C, device=cuda) * 0.2
    rmean, rvar = torch.randn(C, device=cuda) * 0.1, torch.rand(C, device=cuda) + 0.5
    rm0, rv0 = rmean.clone(), rvar.clone()
    eps, mom = 1e-3, 0.03
    mean, invstd = torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
    out = torch.zeros(B, H, W, C, device=cuda, dtype=torch.bfloat16)
    dx = torch.zeros_like(out)
    dgamma, dbeta = torch.full((C,), 7.0, device=cuda), torch.full((C,), -3.0, device=cuda)      # overwritten (accumulate = 0)
    vr, vo, vres, vdy, vdx = view(raw, 8, C), view(out, 0, C), view(res, 0, C) if with_res else NULLV, view(dy, 0, C), view(dx, 0, C)
    nb = l.yp_bn_workspace_bytes(B, H, W, C) + 8 * C + 4096
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    check(l.yp_bn_stats(vr, code, B, eps, mom, mean.data_ptr(), invstd.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), ws.data_ptr(), nb, st))
    check(l.yp_bn_act_apply(vr, vo, vres, code, B, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, st))
    check(l.yp_bn_act_bwd(vr, vdy, vdx, code, B, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, dgamma.data_ptr(),
                          dbeta.data_ptr(), 0, ws.data_ptr(), nb, st))
    torch.cuda.synchronize()
    # torch reference on the same rounded inputs (fp32 math)
    x = raw[..., 8:8 + C].float().permute(0, 3, 1, 2).contiguous().requires_grad_()
    g_, b_ = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    y = torch.nn.functional.batch_norm(x, rm0.clone(), rv0.clone(), g_, b_, True, mom, eps)
    y = torch.nn.functional.silu(y) if act else y
    ref_out = y + (res.float().permute(0, 3, 1, 2) if with_res else 0)
    y.backward(dy.float().permute(0, 3, 1, 2))
    m_ref = x.detach().mean((0, 2, 3))
    v_ref = x.detach().var((0, 2, 3), unbiased=False)
    n = B * H * W
    assert torch.allclose(mean, m_ref, rtol=1e-4, atol=1e-5)
    assert torch.allclose(invstd, 1.0 / torch.sqrt(v_ref + eps), rtol=1e-4)
    assert torch.allclose(rmean, (1 - mom) * rm0 + mom * m_ref, rtol=1e-4, atol=1e-5)
    assert torch.allclose(rvar, (1 - mom) * rv0 + mom * v_ref * n / max(n - 1, 1), rtol=1e-4)
    got_out = out.float().permute(0, 3, 1, 2)
    assert float((got_out - ref_out).abs().max() / ref_out.abs().max()) < 1e-2               # bf16 output rounding
    got_dx = dx.float().permute(0, 3, 1, 2)
    assert float((got_dx - x.grad).norm() / x.grad.norm()) < 1e-2
    assert torch.allclose(dgamma, g_.grad, rtol=2e-3, atol=2e-4 * float(g_.grad.abs().max()))
    assert torch.allclose(dbeta, b_.grad, rtol=2e-3, atol=2e-4 * float(b_.grad.abs().max()))


@pytest.mark.parametrize("C,B,H,W,act,with_res,groups", [(32, 8, 40, 40, 1, False, 2), (64, 4, 20, 24, 1, True, 2), (72, 6, 16, 16, 1, False, 2),
                                                          (256, 16, 20, 20, 1, False, 2), (128, 6, 9, 11, 0, True, 3), (512, 16, 80, 80, 1, False, 2)])
def test_bn_statistics_groups(cuda, C, B, H, W, act, with_res, groups):
    """`groups` consecutive sample sets, each normalised with its own batch statistics in one launch per pass == the module called once per
    set, in order (reference train.py:208,220: model(img), model(img_warp)): outputs, running statistics after BOTH updates, input
    gradients per set, parameter gradients summed over the sets.  C = 72 takes the general kernels (C/8 not a power of two), (512, 16, 80,
    80) the 256-thread fold."""
    torch.manual_seed(C + H + groups)
    l, st = lib(), _hip.stream_ptr()
    code = _hip.YP_BF16
    raw = (torch.randn(B, H, W, C + 8, device=cuda) * 1.7 + 0.3)
    raw[B // groups:] = raw[B // groups:] * 0.5 - 1.0            # (the sets have visibly different statistics)
    raw = raw.to(torch.bfloat16)
    res = torch.randn(B, H, W, C, device=cuda).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, C, device=cuda) * 0.1).to(torch.bfloat16)
    gamma, beta = torch.rand(C, device=cuda) + 0.5, torch.randn(C, device=cuda) * 0.2
    rmean, rvar = torch.randn(C, device=cuda) * 0.1, torch.rand(C, device=cuda) + 0.5
    rm_ref, rv_ref = rmean.clone(), rvar.clone()
    eps, mom = 1e-3, 0.03
    mean, invstd = torch.zeros(groups, C, device=cuda), torch.zeros(groups, C, device=cuda)
    out = torch.zeros(B, H, W, C, device=cuda, dtype=torch.bfloat16)
    dx = torch.zeros_like(out)
    dgamma, dbeta = torch.full((C,), 7.0, device=cuda), torch.full((C,), -3.0, device=cuda)
    vr, vo, vres, vdy, vdx = view(raw, 8, C), view(out, 0, C), view(res, 0, C) if with_res else NULLV, view(dy, 0, C), view(dx, 0, C)
    nb = l.yp_bn_workspace_bytes(B, H, W, C) + 8 * groups * C + 4096
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    check(l.yp_bn_stats_grouped(vr, code, B, groups, eps, mom, mean.data_ptr(), invstd.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), ws.data_ptr(), nb, st))
    check(l.yp_bn_act_apply_grouped(vr, vo, vres, code, B, groups, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, st))
    check(l.yp_bn_act_bwd_grouped(vr, vdy, vdx, code, B, groups, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act,
                                  dgamma.data_ptr(), dbeta.data_ptr(), 0, ws.data_ptr(), nb, st))
    torch.cuda.synchronize()
    g_, b_ = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    Bg = B // groups
    for g in range(groups):
        sl = slice(g * Bg, (g + 1) * Bg)
        x = raw[sl, ..., 8:8 + C].float().permute(0, 3, 1, 2).contiguous().requires_grad_()
        y = torch.nn.functional.batch_norm(x, rm_ref, rv_ref, g_, b_, True, mom, eps)          # (updates rm_ref / rv_ref in place, set after set)
        y = torch.nn.functional.silu(y) if act else y
        ref_out = y + (res[sl].float().permute(0, 3, 1, 2) if with_res else 0)
        y.backward(dy[sl].float().permute(0, 3, 1, 2))
        assert torch.allclose(mean[g], x.detach().mean((0, 2, 3)), rtol=1e-4, atol=1e-5)
        assert torch.allclose(invstd[g], 1.0 / torch.sqrt(x.detach().var((0, 2, 3), unbiased=False) + eps), rtol=1e-4)
        got_out = out[sl].float().permute(0, 3, 1, 2)
        assert float((got_out - ref_out.detach()).abs().max() / ref_out.detach().abs().max()) < 1e-2
        got_dx = dx[sl].float().permute(0, 3, 1, 2)
        assert float((got_dx - x.grad).norm() / x.grad.norm()) < 1e-2
    assert torch.allclose(rmean, rm_ref, rtol=1e-4, atol=1e-5)
    assert torch.allclose(rvar, rv_ref, rtol=1e-4)
    assert torch.allclose(dgamma, g_.grad, rtol=2e-3, atol=2e-4 * float(g_.grad.abs().max()))
    assert torch.allclose(dbeta, b_.grad, rtol=2e-3, atol=2e-4 * float(b_.grad.abs().max()))


@pytest.mark.parametrize("C,B,H,W,acc", [(64, 2, 20, 20, 0), (256, 3, 10, 12, 1), (8, 1, 7, 5, 1), (24, 2, 6, 6, 0)])
def test_maxpool5_backward_vector_and_scalar_paths(cuda, C, B, H, W, acc):
    """Gradient of MaxPool2d(5, 1, 2) (SPPF, reference models/common.py:200-214) on bf16 NHWC views: against torch's CPU backward on the
    same rounded tensors (first maximum of the row-major window scan takes the gradient -- bf16 maps hold ties), and the 8-channels-per-
    thread kernels against the one-element-per-thread kernels (a view that starts 2 bytes off a 16-byte boundary takes those): bit for bit."""
    torch.manual_seed(C + H)
    l, st = lib(), _hip.stream_ptr()
    code = _hip.YP_BF16
    nb = l.yp_maxpool5_bwd_workspace_bytes(B, H, W, C)
    x = torch.randn(B, H, W, C + 8, device=cuda).to(torch.bfloat16)
    dy = torch.randn(B, H, W, C, device=cuda).to(torch.bfloat16)
    dx0 = torch.randn(B, H, W, C, device=cuda).to(torch.bfloat16)
    outs = []
    for shift in (0, 1):                 # 1: every tensor lives one element into its allocation -> unaligned -> scalar kernels
        def place(t):
            flat = torch.zeros(t.numel() + 16, dtype=t.dtype, device=cuda)
            flat[shift:shift + t.numel()] = t.reshape(-1)
            return flat, flat[shift:shift + t.numel()].view(t.shape)
        (kx, xs), (kd, ds), (kg, gs) = place(x), place(dy), place(dx0)
        ws = torch.empty(nb + 16, dtype=torch.uint8, device=cuda)
        check(l.yp_maxpool5_bwd(view(xs, 8, C), view(ds, 0, C), view(gs, 0, C), code, B, acc, ws.data_ptr(), nb, st))
        torch.cuda.synchronize()
        outs.append(gs.clone())
    assert torch.equal(outs[0], outs[1])
    xr = x[..., 8:].float().permute(0, 3, 1, 2).cpu().contiguous().requires_grad_()
    y = torch.nn.functional.max_pool2d(xr, 5, 1, 2)
    y.backward(dy.float().permute(0, 3, 1, 2).cpu())
    want = xr.grad.permute(0, 2, 3, 1) + (dx0.float().cpu() if acc else 0.0)
    got = outs[0].float().cpu()
    assert float((got - want).abs().max()) <= 2 ** -7 * float(want.abs().max())      # (one bf16 rounding of the sum)
