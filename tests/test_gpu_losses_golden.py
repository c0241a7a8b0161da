"""The native loss kernels (csrc/losses.hip: yp_detloss, yp_objloss_level, yp_infonce_*, yp_points_sample_*) against the REFERENCE's
values and gradients (tests/golden/losses.npz, captured by importing the reference with its random draws replayed) --
not against the package's own PyTorch formulation."""
import os

import numpy as np
import pytest
import torch

from helpers import NAMES80, rel_err
from yolopoint_amd import models
from yolopoint_amd.utils import utils as U
from yolopoint_amd.utils.loss_functions import ComputeDetectorLoss, ComputeObjectLoss, infonce

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))


def T(k, dev):
    return torch.from_numpy(G[k]).to(dev)


def test_detector_loss_kernel_vs_reference(cuda):
    semi = T("det.semi", cuda).requires_grad_(True)
    loss = ComputeDetectorLoss(cuda)(semi, U.labels2Dto3D(T("det.labels", cuda)), U.getMasks(T("det.mask", cuda), cuda))
    assert type(loss.grad_fn).__name__.startswith("_DetLossNative"), "the native kernel must be the path that ran"
    np.testing.assert_allclose(loss.item(), G["det.loss"], rtol=1e-5)
    loss.backward()
    assert rel_err(semi.grad, G["det.grad_semi"])[1] < 1e-5


@pytest.mark.parametrize("case", ["obj", "obj2"])
def test_object_loss_kernel_vs_reference(cuda, case):
    model = models.Model(names=NAMES80, model_name="YOLOPoint", version="n").to(cuda)
    hyp = (dict(cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0, box=0.05, obj=1.0, cls=0.5, anchor_t=4.0) if case == "obj" else
           dict(cls_pw=0.7, obj_pw=1.3, fl_gamma=0.0, box=0.05, obj=1.0, cls=0.5, anchor_t=4.0, label_smoothing=0.1))
    p = [T(f"obj.p{i}", cuda).requires_grad_(True) for i in range(3)]
    loss, parts = ComputeObjectLoss(model, hyp, cuda)(p, T(f"{case}.targets", cuda))
    assert "Native" in type(loss.grad_fn).__name__
    np.testing.assert_allclose(loss.detach().cpu().numpy(), G[f"{case}.loss"], rtol=2e-5)
    np.testing.assert_allclose(parts.cpu().numpy(), G[f"{case}.parts"], rtol=2e-5, atol=1e-7)
    loss *= 1.0                                    # the reference step scales the loss in place (train.py:238-240): must be allowed
    loss.backward()
    for i, t in enumerate(p):
        assert rel_err(t.grad, G[f"{case}.grad_p{i}"])[1] < 2e-5, i
    if case == "obj":
        loss0, _ = ComputeObjectLoss(model, hyp, cuda)([t.detach() for t in p], torch.zeros((0, 6), device=cuda))
        np.testing.assert_allclose(loss0.cpu().numpy(), G["obj.loss_empty"], rtol=2e-5)


def test_infonce_kernels_vs_reference(cuda):
    """64-D descriptors (the native gather kernels take D % 64 == 0) with the reference's draws replayed: value and both gradients."""
    rs = np.random.RandomState(4)
    pg = torch.Generator().manual_seed(10)
    # channels-innermost memory, as the network emits its descriptor map (the native point-sample kernel's layout)
    d1 = T("nce64.d1", cuda).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    d2 = T("nce64.d2", cuda).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    loss = infonce(d1, d2, T("nce.mask", cuda), T("nce.Hinv", cuda), num_samples_per_image=50, num_masked_non_matches_per_match=20,
                   device=cuda, perm_fn=lambda n: torch.randperm(n, generator=pg), randint_fn=rs.randint)
    assert "InfoNCENative" in type(loss.grad_fn).__name__
    np.testing.assert_allclose(loss.item(), G["nce64.loss"], rtol=1e-5)
    loss.backward()
    assert rel_err(d1.grad, G["nce64.grad_d1"])[1] < 1e-5 and rel_err(d2.grad, G["nce64.grad_d2"])[1] < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("D", [64, 128, 256])
def test_infonce_gather_generations_agree(cuda, monkeypatch, D):
    """The second-generation gathers (csrc/losses.hip: infonce_fwd_grad2 / infonce_bwd_b3 -- 16-byte loads by D/4 lanes, 8 loads in flight,
    per-lane-group streaming softmax) against the first generation (YP_NCE_GEN=1) and against the sums written out in fp64: loss rows,
    softmax weights, both gradients; ragged edge lists (a column nobody drew, one drawn by every row), E not a multiple of the lanes' stride."""
    from yolopoint_amd import _hip
    from yolopoint_amd.utils.loss_functions import infonce_edges
    n, negs, tau = 777, 37, 0.07
    g = torch.Generator().manual_seed(5 + D)
    dab = torch.nn.functional.normalize(torch.randn((2 * n, D), generator=g), dim=1).to(cuda)
    rnd = torch.randint(1, n, (n, negs), generator=g)
    rnd[:, 3] = 11                                       # column 11 is drawn by every row, column 0 by none
    idx, order, offsets = infonce_edges(rnd.to(cuda))
    E = idx.shape[1]
    lib = _hip.lib()
    scale = torch.full((1,), 1.0 / (tau * n), dtype=torch.float32, device=cuda)
    res = {}
    for gen in ("1", "2"):
        monkeypatch.setenv("YP_NCE_GEN", gen)
        w = torch.empty((n, E), dtype=torch.float32, device=cuda)
        rows, lse = torch.empty((n,), dtype=torch.float32, device=cuda), torch.empty((n,), dtype=torch.float32, device=cuda)
        grad, out = torch.empty_like(dab), torch.zeros_like(dab)
        for cap in (0, 7):                                # uncapped, and a grid that walks the rows
            _hip.check(lib.yp_infonce_fwd_grad(dab.data_ptr(), dab.data_ptr() + 4 * n * D, idx.data_ptr(), n, E, D, 1.0 / tau, w.data_ptr(), rows.data_ptr(),
                                               lse.data_ptr(), grad.data_ptr(), None, cap, _hip.stream_ptr()))
            _hip.check(lib.yp_infonce_bwd_db(dab.data_ptr(), order.data_ptr(), offsets.data_ptr(), w.data_ptr(), lse.data_ptr(), n, E, D, scale.data_ptr(),
                                             out.data_ptr() + 4 * n * D, None, cap, _hip.stream_ptr()))
            res[(gen, cap)] = (rows.clone(), w.clone(), grad[:n].clone(), out[n:].clone())
    a, b = dab[:n].double(), dab[n:].double()
    lg = (a[:, None, :] * b[idx.long()]).sum(-1) / tau
    wref = torch.softmax(lg, 1)
    wref[:, 0] -= 1.0
    ref = (torch.logsumexp(lg, 1) - lg[:, 0], wref, (wref[:, :, None] * b[idx.long()]).sum(1),
           torch.zeros((n, D), dtype=torch.float64, device=cuda).index_add_(0, idx.long().flatten(), (wref[:, :, None] * a[:, None, :]).reshape(-1, D)) * float(scale))
    for key, got in res.items():
        for name, x, r in zip(("loss", "w", "dda", "ddb"), got, ref):
            err = float((x.double() - r).abs().max() / r.abs().max())
            assert err < 5e-6, (key, name, err)
    for cap in (0, 7):                                    # a capped grid changes which wave takes a row, not the row's arithmetic
        for x, y in zip(res[("2", 0)], res[("2", cap)]):
            assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [64, 128, 256])
def test_infonce_gathers_over_16_bit_rows(cuda, D):
    """yp_infonce_fwd_grad_h / yp_infonce_bwd_db_h (the gathered rows as bf16: half the gathered bytes; what engine.TrainStep runs for bf16 / fp8
    graphs) -- (a) exact against the fp64 sums over the SAME bf16-rounded rows (the kernels' only freedom is the fp32 summation order: 5e-6),
    (b) against the fp32-row kernels' definition: the difference is the bf16 rounding of the gathered rows and nothing else (loss rows,
    softmax weights and both gradients within 1e-2 of their scale); yp_infonce_rows16 = torch's round-to-nearest-even bf16 cast, bit for bit;
    a capped grid gives the same bits."""
    from yolopoint_amd import _hip
    from yolopoint_amd.utils.loss_functions import infonce_edges
    n, negs, tau = 777, 37, 0.07
    g = torch.Generator().manual_seed(15 + D)
    dab = torch.nn.functional.normalize(torch.randn((2 * n, D), generator=g), dim=1).to(cuda)
    rnd = torch.randint(1, n, (n, negs), generator=g)
    rnd[:, 3] = 11
    idx, order, offsets = infonce_edges(rnd.to(cuda))
    E = idx.shape[1]
    lib = _hip.lib()
    scale = torch.full((1,), 1.0 / (tau * n), dtype=torch.float32, device=cuda)
    rows16 = torch.empty((2 * n, D), dtype=torch.bfloat16, device=cuda)
    _hip.check(lib.yp_infonce_rows16(dab.data_ptr(), 2 * n * D, rows16.data_ptr(), _hip.stream_ptr()))
    assert torch.equal(rows16, dab.bfloat16())
    res = {}
    for cap in (0, 7):
        w = torch.empty((n, E), dtype=torch.float32, device=cuda)
        rows, lse = torch.empty((n,), dtype=torch.float32, device=cuda), torch.empty((n,), dtype=torch.float32, device=cuda)
        grad, out = torch.empty_like(dab), torch.zeros_like(dab)
        _hip.check(lib.yp_infonce_fwd_grad_h(dab.data_ptr(), rows16.data_ptr(), idx.data_ptr(), n, E, D, 1.0 / tau, w.data_ptr(), rows.data_ptr(),
                                             lse.data_ptr(), grad.data_ptr(), None, cap, _hip.stream_ptr()))
        _hip.check(lib.yp_infonce_bwd_db_h(rows16.data_ptr(), order.data_ptr(), offsets.data_ptr(), w.data_ptr(), n, E, D, scale.data_ptr(),
                                           out.data_ptr() + 4 * n * D, None, cap, _hip.stream_ptr()))
        res[cap] = (rows.clone(), w.clone(), grad[:n].clone(), out[n:].clone())
    for x, y in zip(res[0], res[7]):
        assert torch.equal(x, y)

    def sums(a, b, a_gathered):
        lg = (a[:, None, :] * b[idx.long()]).sum(-1) / tau
        wref = torch.softmax(lg, 1)
        wref[:, 0] -= 1.0
        return (torch.logsumexp(lg, 1) - lg[:, 0], wref, (wref[:, :, None] * b[idx.long()]).sum(1),
                torch.zeros((n, D), dtype=torch.float64, device=cuda).index_add_(0, idx.long().flatten(), (wref[:, :, None] * a_gathered[:, None, :]).reshape(-1, D)) * float(scale))
    h = rows16.double()
    exact = sums(dab[:n].double(), h[n:], h[:n])           # anchors' own rows fp32, every GATHERED row bf16 -- what the kernels compute
    full = sums(dab[:n].double(), dab[n:].double(), dab[:n].double())
    for name, x, r, f in zip(("loss", "w", "dda", "ddb"), res[0], exact, full):
        assert float((x.double() - r).abs().max() / r.abs().max()) < 5e-6, name
        assert float((x.double() - f).abs().max() / f.abs().max()) < 1e-2, name
