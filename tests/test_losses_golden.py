"""CPU: the PyTorch statements of the training losses reproduce the reference's values and gradients (tests/golden/losses.npz,
captured by importing the reference with seeded / injected random draws): the detector loss and InfoNCE formulations of the package
(used for CPU tensors) and the oracle's statement of the YOLO object loss / CIoU / target assignment (oracle/loss_oracle.py -- the
product computes those on the device only; tests/test_gpu_losses_golden.py pins the kernels to the same file)."""
import os

import numpy as np
import pytest
import torch

from helpers import NAMES80
from yolopoint_amd import models
from yolopoint_amd.utils import utils as U
from oracle import loss_oracle
from yolopoint_amd.utils.loss_functions import ComputeDetectorLoss, infonce
from yolopoint_amd.utils.metrics_yolo import box_iou

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
T = lambda k: torch.from_numpy(G[k])


def test_detector_loss():
    tgt = U.labels2Dto3D(T("det.labels"))
    m3 = U.getMasks(T("det.mask"), "cpu")
    loss = ComputeDetectorLoss("cpu")(T("det.semi"), tgt, m3)
    np.testing.assert_allclose(loss.numpy(), G["det.loss"], rtol=1e-5)


def test_ciou():
    np.testing.assert_allclose(loss_oracle.ciou(T("iou.b1"), T("iou.b2")).numpy(), G["iou.ciou"][:, 0], rtol=1e-5, atol=1e-6)
    a = torch.tensor([[0., 0., 2., 2.]]); b = torch.tensor([[1., 1., 3., 3.], [0., 0., 2., 2.]])
    assert torch.allclose(box_iou(a, b), torch.tensor([[1 / 7, 1.0]]), atol=1e-6)


HYPS = {"obj": dict(cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0, box=0.05, obj=1.0, cls=0.5, anchor_t=4.0),
        "obj2": dict(cls_pw=0.7, obj_pw=1.3, fl_gamma=0.0, box=0.05, obj=1.0, cls=0.5, anchor_t=4.0, label_smoothing=0.1)}


@pytest.mark.parametrize("case", ["obj", "obj2"])
def test_object_loss_and_gradient(case):
    """oracle/loss_oracle.py::object_loss (value, parts, gradient w.r.t. all three level tensors; obj2 has duplicated cell claims, label
    smoothing and positive weights) against the reference."""
    model = models.Model(names=NAMES80, model_name="YOLOPoint", version="n")
    anchors = model.model.Detect.anchors
    p = [T(f"obj.p{i}").requires_grad_(True) for i in range(3)]
    loss, parts = loss_oracle.object_loss(p, T(f"{case}.targets"), anchors, 80, HYPS[case])
    np.testing.assert_allclose(loss.detach().numpy(), G[f"{case}.loss"], rtol=1e-5)
    np.testing.assert_allclose(parts.numpy(), G[f"{case}.parts"], rtol=1e-5)
    loss.backward()
    for i, t in enumerate(p):
        np.testing.assert_allclose(t.grad.numpy(), G[f"{case}.grad_p{i}"], rtol=1e-4, atol=1e-8)
    if case == "obj":
        loss0, _ = loss_oracle.object_loss([t.detach() for t in p], torch.zeros((0, 6)), anchors, 80, HYPS[case])
        np.testing.assert_allclose(loss0.numpy(), G["obj.loss_empty"], rtol=1e-5)


def test_detector_loss_gradient():
    semi = T("det.semi").requires_grad_(True)
    ComputeDetectorLoss("cpu")(semi, U.labels2Dto3D(T("det.labels")), U.getMasks(T("det.mask"), "cpu")).backward()
    np.testing.assert_allclose(semi.grad.numpy(), G["det.grad_semi"], rtol=1e-4, atol=1e-9)


def test_infonce_with_injected_draws():
    rs = np.random.RandomState(3)
    pg = torch.Generator().manual_seed(9)
    d1 = T("nce.d1").requires_grad_(True)
    loss = infonce(d1, T("nce.d2"), T("nce.mask"), T("nce.Hinv"), num_samples_per_image=40, num_masked_non_matches_per_match=12,
                   device="cpu", perm_fn=lambda n: torch.randperm(n, generator=pg), randint_fn=rs.randint)
    np.testing.assert_allclose(loss.detach().numpy(), G["nce.loss"], rtol=1e-5)
    loss.backward()
    assert torch.isfinite(d1.grad).all() and float(d1.grad.abs().sum()) > 0


def test_infonce_64d_value_and_gradients():
    rs = np.random.RandomState(4)
    pg = torch.Generator().manual_seed(10)
    d1, d2 = T("nce64.d1").requires_grad_(True), T("nce64.d2").requires_grad_(True)
    loss = infonce(d1, d2, T("nce.mask"), T("nce.Hinv"), num_samples_per_image=50, num_masked_non_matches_per_match=20,
                   device="cpu", perm_fn=lambda n: torch.randperm(n, generator=pg), randint_fn=rs.randint)
    np.testing.assert_allclose(loss.detach().numpy(), G["nce64.loss"], rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(d1.grad.numpy(), G["nce64.grad_d1"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(d2.grad.numpy(), G["nce64.grad_d2"], rtol=1e-4, atol=1e-8)
