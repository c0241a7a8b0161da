"""GPU parity of the homography-adaptation export (reference export_homography.py:23-178) against the golden vectors captured
from the imported reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import make_model
from oracle import net_oracle, postproc_oracle as po
from yolopoint_amd.export_homography import HomographyExporter, combine_heatmaps, homographic_export
from yolopoint_amd.utils import utils as U

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


class _PlantedHead(torch.nn.Module):
    """Stands in for the network: returns planted keypoint logits for the views it is given."""
    def __init__(self, semi):
        super().__init__()
        self.semi = semi

    def forward(self, x):
        assert x.shape[0] == self.semi.shape[0]
        return {"semi": self.semi.to(x.device)}


def homographies(n, seed):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 3, 3), dtype=np.float32)
    for i in range(n):
        a, s = rng.uniform(-0.5, 0.5), 1.0 + rng.uniform(-0.25, 0.25)
        out[i] = [[s * np.cos(a), -s * np.sin(a), rng.uniform(-0.2, 0.2)], [s * np.sin(a), s * np.cos(a), rng.uniform(-0.2, 0.2)],
                  [rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15), 1.0]]
    out[0] = np.eye(3)
    return out


def valid_masks(homs, H, W):
    m = po.warp_image_batch(np.ones((len(homs), 1, H, W), np.float32), homs, mode="nearest")
    m[:, :, :1, :] = 0; m[:, :, -1:, :] = 0; m[:, :, :, :1] = 0; m[:, :, :, -1:] = 0
    return m


@pytest.mark.parametrize("i", [0, 1, 2])
def test_export_matches_reference_golden(cuda, i):
    g = np.load(os.path.join(G, "export.npz"))
    N, Hc, Wc, thr, r, top_k = g[f"ha{i}.cfg"]
    N, Hc, Wc, r, top_k = int(N), int(Hc), int(Wc), int(r), int(top_k)
    semi = torch.from_numpy(g[f"ha{i}.semi"])
    mask = torch.from_numpy(g[f"ha{i}.valid_mask"].astype(np.float32))
    inv = torch.from_numpy(g[f"ha{i}.inv_homographies"])
    heat = U.flattenDetection(semi.to(cuda))
    agg = combine_heatmaps(heat, mask, inv).cpu().numpy()
    ref = g[f"ha{i}.agg"]
    assert np.array_equal(np.isnan(agg), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(agg[ok], ref[ok], rtol=1e-5, atol=1e-5)       # fp32 source-coordinate rounding (see the oracle test)
    exp = HomographyExporter(_PlantedHead(semi), cuda, dict(nms=r, top_k=top_k, detection_threshold=thr))
    sample = {"image": torch.zeros(1, N, 3, Hc * 8, Wc * 8), "valid_mask": mask.view(1, N, Hc * 8, Wc * 8), "inv_homographies": inv[None]}
    pts = exp.export_sample(sample)
    rp = g[f"ha{i}.pts"]
    assert pts.shape == rp.shape and pts.dtype == np.float64
    assert np.array_equal(pts[:, :2], rp[:, :2])
    np.testing.assert_allclose(pts[:, 2], rp[:, 2], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,H,W", [(100, 240, 320), (7, 96, 64), (1, 64, 64)])
def test_combine_against_oracle(cuda, N, H, W):
    rng = np.random.default_rng(N + H)
    heat = rng.random((N, 1, H, W), dtype=np.float32) ** 4
    homs = homographies(N, N)
    inv = np.linalg.inv(homs.astype(np.float64)).astype(np.float32)
    mask = valid_masks(homs, H, W)
    ref_h = po.warp_image_batch(heat * mask, inv).sum(axis=0, dtype=np.float32)[0]
    ref_m = po.warp_image_batch(mask, inv).sum(axis=0, dtype=np.float32)[0]
    got, cover = combine_heatmaps(torch.from_numpy(heat).to(cuda), torch.from_numpy(mask).to(cuda), torch.from_numpy(inv).to(cuda), want_cover=True)
    np.testing.assert_allclose(cover.cpu().numpy(), ref_m, rtol=1e-5, atol=2e-5 * max(1, N) ** 0.5)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = ref_h / ref_m
    got = got.cpu().numpy()
    solid = ref_m > 1e-3                      # (a denominator of ~0 amplifies the coordinate rounding without bound)
    np.testing.assert_allclose(got[solid], ref[solid], rtol=2e-5, atol=2e-5)


def test_export_end_to_end_with_network(cuda, tmp_path):
    """The real network on warped views: the exporter's points == the oracle's decode of the same aggregated map, and the
    aggregated map == the oracle's aggregation of the network's own head output; the .npz holds {'pts': [n,3]}."""
    N, S = 6, 128
    m, _ = make_model("n", 5, dtype="f16")
    m = m.to(cuda).eval()
    base = net_oracle.synth_image(1, 3, S, S, 3)
    homs = homographies(N, 11)
    inv = np.linalg.inv(homs.astype(np.float64)).astype(np.float32)
    views = torch.from_numpy(po.warp_image_batch(np.repeat(base.numpy(), N, axis=0), homs))
    mask = valid_masks(homs, S, S)
    sample = {"name": ["img0"], "image": views[None], "valid_mask": torch.from_numpy(mask).view(1, N, S, S), "inv_homographies": torch.from_numpy(inv)[None]}
    exp = HomographyExporter(m, cuda, dict(nms=4, top_k=0, detection_threshold=0.02))
    agg = exp.aggregate(views, torch.from_numpy(mask), torch.from_numpy(inv))
    with torch.no_grad():
        semi = m(views.to(cuda))["semi"].float().cpu().numpy()
    ref_agg, _ = po.homography_adaptation(semi, mask, inv, 0.02, 4, 0)
    a = agg.cpu().numpy()
    ok = ~np.isnan(ref_agg)
    assert np.array_equal(np.isnan(a), ~ok)
    np.testing.assert_allclose(a[ok], ref_agg[ok], rtol=2e-5, atol=2e-5)
    pts = exp.export_sample(sample)
    ref_pts = po.get_pts_from_heatmap(a, 0.02, 4).transpose()
    assert pts.shape == ref_pts.shape and pts.shape[0] > 0
    assert np.array_equal(pts, ref_pts.astype(np.float32).astype(np.float64)) or np.allclose(pts, ref_pts, rtol=1e-7, atol=0)
    cfg = {"data": {"dataset": "synthetic", "homography_adaptation": {"nms": 4, "top_k": 50, "detection_threshold": 0.02}}}
    paths = homographic_export(cfg, [sample], m, tmp_path, export_task="train", device=cuda)
    z = np.load(paths[0])
    assert list(z.keys()) == ["pts"] and z["pts"].shape == (min(50, pts.shape[0]), 3)
    assert np.array_equal(z["pts"], pts[:50])
