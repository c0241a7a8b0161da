"""CPU: the PyTorch-autograd training losses of the package reproduce the reference's values
(tests/golden/losses.npz, captured by importing the reference with seeded / injected random draws)."""
import os

import numpy as np
import pytest
import torch

from helpers import NAMES80
from yolopoint_amd import models
from yolopoint_amd.utils import utils as U
from yolopoint_amd.utils.loss_functions import ComputeDetectorLoss, ComputeObjectLoss, infonce
from yolopoint_amd.utils.metrics_yolo import bbox_iou, box_iou

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
T = lambda k: torch.from_numpy(G[k])


def test_detector_loss():
    tgt = U.labels2Dto3D(T("det.labels"))
    m3 = U.getMasks(T("det.mask"), "cpu")
    loss = ComputeDetectorLoss("cpu")(T("det.semi"), tgt, m3)
    np.testing.assert_allclose(loss.numpy(), G["det.loss"], rtol=1e-5)


def test_ciou():
    np.testing.assert_allclose(bbox_iou(T("iou.b1"), T("iou.b2"), CIoU=True).numpy(), G["iou.ciou"], rtol=1e-5, atol=1e-6)
    a = torch.tensor([[0., 0., 2., 2.]]); b = torch.tensor([[1., 1., 3., 3.], [0., 0., 2., 2.]])
    assert torch.allclose(box_iou(a, b), torch.tensor([[1 / 7, 1.0]]), atol=1e-6)


def test_object_loss_and_gradient():
    model = models.Model(names=NAMES80, model_name="YOLOPoint", version="n")
    hyp = dict(cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0, box=0.05, obj=1.0, cls=0.5, anchor_t=4.0)
    p = [T(f"obj.p{i}").requires_grad_(True) for i in range(3)]
    crit = ComputeObjectLoss(model, hyp, "cpu")
    loss, parts = crit(p, T("obj.targets"))
    np.testing.assert_allclose(loss.detach().numpy(), G["obj.loss"], rtol=1e-5)
    np.testing.assert_allclose(parts.numpy(), G["obj.parts"], rtol=1e-5)
    loss.backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in p)
    loss0, _ = crit([t.detach() for t in p], torch.zeros((0, 6)))
    np.testing.assert_allclose(loss0.numpy(), G["obj.loss_empty"], rtol=1e-5)


def test_infonce_with_injected_draws():
    rs = np.random.RandomState(3)
    pg = torch.Generator().manual_seed(9)
    d1 = T("nce.d1").requires_grad_(True)
    loss = infonce(d1, T("nce.d2"), T("nce.mask"), T("nce.Hinv"), num_samples_per_image=40, num_masked_non_matches_per_match=12,
                   device="cpu", perm_fn=lambda n: torch.randperm(n, generator=pg), randint_fn=rs.randint)
    np.testing.assert_allclose(loss.detach().numpy(), G["nce.loss"], rtol=1e-5)
    loss.backward()
    assert torch.isfinite(d1.grad).all() and float(d1.grad.abs().sum()) > 0
