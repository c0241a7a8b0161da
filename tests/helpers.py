"""Shared test helpers: seeded synthetic checkpoints in the reference layout + error metrics."""
import numpy as np
import torch

from oracle import net_oracle
from yolopoint_amd import models

from yolopoint_amd.utils.synthetic import (NAMES80, layout_of, make_model, planted_heatmap, planted_predictions,  # noqa: F401
                                           planted_descriptors, tracking_sequence)


def block_state(module, seed, prefix=""):
    """Randomise a single block's parameters/buffers (same recipe as synth_state_dict); returns sd."""
    layout = [(prefix + k, tuple(v.shape)) for k, v in module.state_dict().items()]
    sd = net_oracle.synth_state_dict(layout, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()}, strict=True)
    return sd


def rel_err(a, b):
    """max |a-b| / max |b|  and  ||a-b||_2 / ||b||_2  (a: test, b: reference)."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / den), float((a - b).norm() / b.norm().clamp_min(1e-30))


# tolerances (max-abs error relative to the tensor's max magnitude) per compute dtype
TOL = {"f32": 1e-4, "f16": 4e-3, "bf16": 3e-2}


# ---------------------------------------------------------------------------------------------
# planted post-processing inputs (random-init heads give degenerate workloads)
# ---------------------------------------------------------------------------------------------


def check_grad_sketch(G, tag, name, grad, rel):
    """Compare a gradient tensor with its committed sketch (tests/golden/backward.npz, make_golden.grad_sketch): L2 norm and 8 seeded
    +-1 projections (a projection of an error vector e is ~|e|_2, so the bar on each is 4 * rel * |g_ref|_2), plus the whole tensor
    where the file holds it.  Returns the worst normalised deviation."""
    g = np.asarray(grad.detach().double().cpu()).ravel()
    nref = float(G[f"{tag}.norm.{name}"])
    den = max(nref, 1e-30)
    worst = abs(np.sqrt((g * g).sum()) - nref) / den
    assert worst < rel, (tag, name, "norm", worst)
    proj = net_oracle.sign_projections(name, g.size, 8) @ g
    dev = np.abs(proj - G[f"{tag}.proj.{name}"]).max() / den
    assert dev < 4 * rel, (tag, name, "projection", dev)
    worst = max(worst, dev / 4)
    key = f"{tag}.full.{name}"
    if key in G:
        e = np.sqrt(((g - G[key].astype(np.float64).ravel()) ** 2).sum()) / den
        assert e < rel, (tag, name, "full", e)
        worst = max(worst, e)
    return worst


def argmax_mismatches(got, ref, tie_tol):
    """Cells whose channel argmax differs between `got` and `ref` ([B,C,H,W]).  Every such cell must be a TIE of the reference within
    `tie_tol` (absolute margin between the reference's value at its own argmax and at the other candidate): fp32 sums evaluated in
    a different order can only flip an argmax where the two logits are equal to rounding.  Returns (mismatching cells, largest margin)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    ia, ib = got.argmax(1, keepdim=True), ref.argmax(1, keepdim=True)
    bad = ia != ib
    if not bool(bad.any()):
        return 0, 0.0
    margin = (ref.gather(1, ib) - ref.gather(1, ia))[bad]
    assert float(margin.max()) <= tie_tol, f"argmax differs on a non-tie: margin {float(margin.max()):.3e} > {tie_tol:.1e}"
    return int(bad.sum()), float(margin.max())


half_storage_oracle, fused_state_dict = net_oracle.half_storage, net_oracle.fused_state_dict          # the 16-bit floor of the network (oracle/net_oracle.py)
