"""BASELINE.json configs[0]: YOLOPoint-nano, one 640x640 synthetic image, CPU forward + box NMS + keypoint decode / NMS -- the
reference's own CPU-runnable case, run here through the oracle (test infrastructure; no GPU).  Checks the plumbing end to end:
shapes, value ranges, and that the post-processing of planted head outputs returns what was planted."""
import numpy as np
import torch

from helpers import NAMES80, layout_of, planted_heatmap, planted_predictions
from oracle import net_oracle, postproc_oracle as po
from yolopoint_amd import models


def test_config0_nano_640_cpu_forward_and_postprocess():
    torch.set_num_threads(8)
    sd = net_oracle.synth_state_dict(layout_of(models.Model(names=NAMES80, version="n")), 1234)
    x = net_oracle.synth_image(1, 3, 640, 640, 1234)
    with torch.no_grad():
        o = net_oracle.yolopoint_forward(sd, x, "n")
    pred, xs = o["objects"]
    assert o["semi"].shape == (1, 65, 80, 80) and o["desc"].shape == (1, 64, 80, 80) and pred.shape == (1, 25200, 85)
    assert [tuple(t.shape) for t in xs] == [(1, 3, 80, 80, 85), (1, 3, 40, 40, 85), (1, 3, 20, 20, 85)]
    np.testing.assert_allclose(o["desc"].norm(dim=1).numpy(), 1.0, rtol=1e-5)
    heat = po.flatten_detection(o["semi"].numpy())
    assert heat.shape == (1, 1, 640, 640) and 0.0 <= heat.min() and heat.max() <= 1.0
    # keypoint decode + NMS and box NMS on the network's own outputs (whatever the seed gives) ...
    pts = po.get_pts_from_heatmap(heat[0, 0], 0.015, 4)
    assert pts.shape[0] == 3 and (pts[0] >= 4).all() and (pts[0] < 636).all()
    dets = po.non_max_suppression(pred.numpy(), 0.25, 0.45, agnostic=True, multi_label=True, max_det=300)
    assert len(dets) == 1 and dets[0].shape[1] == 6
    # ... and on planted head outputs, where the answer is known
    ph = planted_heatmap(640, 640, 1000, 3)
    kp = po.get_pts_from_heatmap(ph, 0.015, 4)
    assert 300 < kp.shape[1] <= 1000 and np.all(np.diff(kp[2]) <= 0)
    pp = planted_predictions(1, 25200, 80, 2000, 3)
    d = po.non_max_suppression(pp, 0.25, 0.45, agnostic=True, multi_label=True, max_det=300)[0]
    assert 20 < d.shape[0] <= 300 and np.all(np.diff(d[:, 4]) <= 0)
