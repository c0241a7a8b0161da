"""CPU: PyTorch autograd through the oracle reproduces the reference's loss.backward() (train.py:245) -- every parameter
gradient and the input gradient of YOLOPoint-s / -n against tests/golden/backward.npz (the file holds
norm + 8 projections per tensor and the small tensors in full, see make_golden.grad_sketch)."""
import os

import numpy as np
import pytest
import torch

from helpers import NAMES80, layout_of, check_grad_sketch
from oracle import net_oracle
from yolopoint_amd import models

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "backward.npz"))
CASES = {"s64": ("s", 2, 64, 31), "n128": ("n", 3, 128, 32)}


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_autograd_matches_reference_backward(tag):
    v, B, S, seed = CASES[tag]
    sd = net_oracle.synth_state_dict(layout_of(models.Model(names=NAMES80, version=v)), seed)
    leaf = {k: (t.clone().requires_grad_(True) if t.dtype.is_floating_point and "running" not in k else t.clone()) for k, t in sd.items()}
    x = net_oracle.synth_image(B, 3, S, S, seed).requires_grad_(True)
    o = net_oracle.yolopoint_forward(leaf, x, v, training=True, stats={})
    loss = net_oracle.projected_loss(o, net_oracle.output_projections(o, seed))
    assert abs(float(loss) - float(G[f"{tag}.loss"])) <= 1e-4 * abs(float(G[f"{tag}.loss"])) + 1e-3
    loss.backward()
    names = [k[len(tag) + 6:] for k in G.files if k.startswith(tag + ".norm.")]
    assert len(names) == 216 and "input" in names              # 215 parameters + the input
    worst = max(check_grad_sketch(G, tag, n, (x.grad if n == "input" else leaf[n].grad), 5e-4) for n in names)
    print(tag, "worst normalised deviation from the reference gradients:", worst)
