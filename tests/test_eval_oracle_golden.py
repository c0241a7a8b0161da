"""CPU: the metric definitions (repeatability, TP matching, AP) of oracle/eval_oracle.py against the reference's values."""
import os

import numpy as np

from oracle import eval_oracle as eo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval.npz"))


def test_repeatability():
    data = dict(image=np.zeros((3, 96, 128), np.float32), homography=G["rep.H"], inv_homography=G["rep.Hinv"], prob=G["rep.pts"].copy(),
                warped_prob=G["rep.wpts"].copy())
    rep, err = eo.compute_repeatability(data, keep_k_points=300, distance_thresh=3)
    assert abs(rep - G["rep.out"][0]) < 1e-9 and abs(err - G["rep.out"][1]) < 1e-6
    assert 0.3 < rep < 1.0
    empty = dict(data, prob=np.zeros((0, 3)), warped_prob=np.zeros((0, 3)))
    assert eo.compute_repeatability(empty)[0] == 0


def test_tp_matching_and_ap():
    iouv = np.linspace(0.5, 0.95, 10).astype(np.float32)
    correct = eo.process_batch(G["ap.det"], G["ap.labels"], iouv)
    assert np.array_equal(correct, G["ap.correct"])
    ap, cls = eo.ap_per_class(correct, G["ap.det"][:, 4], G["ap.det"][:, 5], G["ap.labels"][:, 0])
    np.testing.assert_allclose(ap, G["ap.ap"], atol=1e-12)
    assert np.array_equal(cls, G["ap.cls"])
    assert 0.0 < ap[:, 0].mean() <= 1.0
