#!/usr/bin/env python3
"""What a training step does when one image of the batch has NO valid cell under the warp (InfoNCE pool size 0): loss terms and gradient
finiteness, native stage, both prepare forms.  python tools/probe/empty_pool_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch

dev = torch.device("cuda:0")
for sync in ("0", "1"):
    os.environ["YP_PREPARE_SYNC"] = sync
    m, _ = make_model("n", 5, dtype="bf16")
    m = m.to(dev).train()
    step = TrainStep(m, dev, img_size=128)
    step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
    batch = synthetic_batch(2, 128, dev, 3)
    batch['warped_valid_mask'][1] = 0.0
    try:
        loss = step(batch)
        torch.cuda.synchronize()
        terms = step.last_loss_terms.tolist() if getattr(step, "last_loss_terms", None) is not None else None
        finite = all(torch.isfinite(p.grad).all().item() for p in m.parameters() if p.grad is not None)
        print(f"sync={sync}: loss {float(loss):.5f} terms {terms} gradients finite {finite}")
    except Exception as e:
        print(f"sync={sync}: {type(e).__name__}: {str(e)[:300]}")
