"""GPU-resident frame pipeline (yolopoint_amd/frontend.py, reference demo.py:125-230) against the oracle's restatement of the
same post-processing chain, fed with the SAME head outputs (so index selection must agree exactly), and the box-mask
keypoint filter kernel against numpy's slice-painting on planted boxes."""
import numpy as np
import pytest
import torch

from helpers import make_model
from oracle import net_oracle, postproc_oracle
from oracle import postproc_oracle as po
from yolopoint_amd import _hip
from yolopoint_amd.frontend import YoloPointFrontend

pytestmark = pytest.mark.gpu


def test_pts_box_filter_matches_numpy_slice_painting(cuda):
    H, W = 96, 128
    rng = np.random.default_rng(3)
    xs, ys = rng.integers(0, W, 700), rng.integers(0, H, 700)
    pts = np.stack([xs, ys, rng.random(700)]).astype(np.float32)                       # [3,N]
    boxes = np.array([[10.4, 5.5, 40.6, 30.5, .9, 0], [-7.2, 50.0, 20.0, 200.0, .8, 1], [100.5, -3.0, 127.49, 20.5, .7, 2],
                      [60.0, 60.0, 60.4, 90.0, .6, 3], [-300.0, -300.0, 5.0, 5.0, .5, 0], [70.0, 40.0, 65.0, 80.0, .4, 0]], dtype=np.float32)
    ref = postproc_oracle.filter_points(boxes, pts.astype(np.float64), H, W)
    p = torch.from_numpy(np.ascontiguousarray(pts.T)).to(cuda)
    b = torch.from_numpy(boxes).to(cuda)
    out, cnt = torch.empty_like(p), torch.zeros(1, dtype=torch.int32, device=cuda)
    _hip.check(_hip.lib().yp_pts_box_filter(p.data_ptr(), None, p.shape[0], b.data_ptr(), None, b.shape[0], 6, H, W, out.data_ptr(), cnt.data_ptr(), _hip.stream_ptr()))
    n = int(cnt.item())
    assert 0 < n < 700 and n == ref.shape[1]
    np.testing.assert_array_equal(out[:n].cpu().numpy().T, ref.astype(np.float32))
    # no boxes: everything is kept, in order
    _hip.check(_hip.lib().yp_pts_box_filter(p.data_ptr(), None, p.shape[0], None, None, 0, 6, H, W, out.data_ptr(), cnt.data_ptr(), _hip.stream_ptr()))
    assert int(cnt.item()) == 700 and torch.equal(out, p)


def test_pts_box_filter_two_pass_path(cuda):
    """More than 1024 points and 16 boxes: the box tests run on all CUs first (verdicts parked in the output buffer), then the
    order-keeping compaction -- 9 000 points (three 4096-point compaction passes), 40 boxes, counts read from the device, capacity larger
    than the count; exact list and order against the reference's slice painting."""
    H, W = 480, 640
    rng = np.random.default_rng(11)
    n, cap = 9000, 12000
    pts = np.stack([rng.integers(0, W, cap), rng.integers(0, H, cap), rng.random(cap)]).astype(np.float32)
    c = rng.uniform([0, 0], [W, H], (40, 2))
    wh = rng.uniform(10, 120, (40, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2, rng.random((40, 1)), np.zeros((40, 1))], 1).astype(np.float32)
    boxes[3, :4] = [-50.5, 100.2, 30.7, 900.0]                                          # negative / out-of-range bounds (numpy slice semantics)
    ref = postproc_oracle.filter_points(boxes[:37], pts[:, :n].astype(np.float64), H, W)
    p = torch.from_numpy(np.ascontiguousarray(pts.T)).to(cuda)
    b = torch.from_numpy(boxes).to(cuda)
    out, cnt = torch.full_like(p, -1.0), torch.zeros(1, dtype=torch.int32, device=cuda)
    n_dev, nb_dev = torch.tensor([n], dtype=torch.int32, device=cuda), torch.tensor([37], dtype=torch.int32, device=cuda)
    _hip.check(_hip.lib().yp_pts_box_filter(p.data_ptr(), n_dev.data_ptr(), cap, b.data_ptr(), nb_dev.data_ptr(), 40, 6, H, W, out.data_ptr(), cnt.data_ptr(),
                                            _hip.stream_ptr()))
    k = int(cnt.item())
    assert 0 < k < n and k == ref.shape[1]
    np.testing.assert_array_equal(out[:k].cpu().numpy().T, ref.astype(np.float32))


@pytest.mark.parametrize("filter_pts", [True, False])
def test_frontend_matches_oracle_postprocessing(cuda, filter_pts):
    m, _ = make_model("n", 17, dtype="f32")
    m = m.to(cuda).eval()
    x = net_oracle.synth_image(1, 3, 96, 128, 4).to(cuda)
    with torch.no_grad():
        outs = m(x)
    pred = outs["objects"][0][0].float()
    thr, max_det = 0.5, 10        # the seeded random heads saturate: keep a handful of boxes so that they do not cover the whole frame
    fe = YoloPointFrontend(m, cuda, yolo_config=dict(conf_thres_box=thr, iou_thres_box=0.45, max_det=max_det), filter_pts=filter_pts)
    r = fe.process_tensor(x)
    ref_pts, ref_desc, ref_boxes = postproc_oracle.frontend_postprocess(
        outs["semi"][0].cpu().numpy(), outs["desc"][0].cpu().numpy(), pred.cpu().numpy(), 0.015, 4, 4, thr, 0.45, max_det, filter_pts)
    boxes = r["boxes"].cpu().numpy()
    assert boxes.shape == ref_boxes.shape and boxes.shape[0] > 3
    np.testing.assert_allclose(boxes, ref_boxes, rtol=1e-5, atol=1e-4)
    pts = r["pts"].cpu().numpy().T
    assert pts.shape == ref_pts.shape and pts.shape[1] > 20
    np.testing.assert_array_equal(pts[:2], ref_pts[:2].astype(np.float32))      # same points, same order
    np.testing.assert_allclose(pts[2], ref_pts[2], rtol=1e-5)
    np.testing.assert_allclose(r["desc"].cpu().numpy(), ref_desc, atol=2e-5)
    assert r["n_before_filter"] >= pts.shape[1]      # (whether these random-weight boxes cover a keypoint is up to the seed; the
                                                     #  planted-box test above exercises the removal itself)


def test_async_keypoint_nms_and_frontend_fallback(cuda):
    """yp_kp_nms_async enqueues a fixed number of fix-point rounds without a host sync and reports the candidates left undecided:
    with enough rounds it equals yp_kp_nms (and the oracle); with too few the counter is non-zero and the front end falls back."""
    from helpers import planted_heatmap
    from yolopoint_amd.utils._ws import workspace
    H, W, r = 160, 192, 4
    heat_np = planted_heatmap(H, W, 120, 5)
    ref = postproc_oracle.get_pts_from_heatmap(heat_np, 0.015, r)                     # [3,n]
    heat = torch.from_numpy(heat_np).to(cuda).contiguous()
    l, st = _hip.lib(), _hip.stream_ptr()
    max_out = (H // (r + 1) + 1) * (W // (r + 1) + 1)
    ws = workspace(cuda, l.yp_kp_nms_workspace_bytes(1, H, W), "kp_nms_test")
    res = {}
    for rounds in (1, 32):
        out = torch.zeros(max_out, 3, device=cuda)
        cnt = torch.zeros(2, dtype=torch.int32, device=cuda)
        _hip.check(l.yp_kp_nms_async(heat.data_ptr(), 1, H, W, 0.015, r, 4, out.data_ptr(), cnt.data_ptr(), max_out, ws.data_ptr(), ws.numel(), rounds,
                                     cnt[1:2].data_ptr(), st))
        n, undecided = cnt.cpu().tolist()
        res[rounds] = (out[:n].cpu().numpy().T, undecided)
    assert res[1][1] > 0                                   # one round cannot resolve clusters: reported, not silently wrong
    assert res[32][1] == 0
    np.testing.assert_array_equal(res[32][0].astype(np.float64)[:2], ref[:2])
    np.testing.assert_allclose(res[32][0][2], ref[2], rtol=1e-6)
    # front end: too few rounds -> the synchronising variant takes over, results identical
    m, _ = make_model("n", 17, dtype="f32")
    m = m.to(cuda).eval()
    x = net_oracle.synth_image(1, 3, 96, 128, 4).to(cuda)
    fe = YoloPointFrontend(m, cuda, yolo_config=dict(conf_thres_box=0.5, iou_thres_box=0.45, max_det=10))
    a = fe.process_tensor(x)
    fe.NMS_ROUNDS = 1
    b = fe.process_tensor(x)
    assert torch.equal(a["pts"], b["pts"]) and torch.equal(a["desc"], b["desc"]) and torch.equal(a["boxes"], b["boxes"])


def test_frontend_process_img_reference_formats(cuda):
    m, _ = make_model("n", 17, dtype="f16")
    m = m.to(cuda).eval()
    img = (net_oracle.synth_image(1, 3, 101, 140, 9)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)    # not a multiple of 32: centre crop
    fe = YoloPointFrontend(m, cuda, yolo_config=dict(conf_thres_box=0.9), filter_pts=True)
    pts, desc, boxes = fe.process_img(img)
    assert pts.dtype == np.float64 and pts.shape[0] == 3 and desc.shape == (m.model.ConvDesc.out_channels, pts.shape[1])
    assert isinstance(boxes, list) and boxes[0].shape[1] == 6
    assert pts[0].min() >= 6 + 4 and pts[1].min() >= 3 + 4                    # crop offsets (ceil(12/2), ceil(5/2)) + border
    np.testing.assert_allclose(np.linalg.norm(desc, axis=0), 1.0, atol=1e-4)
    with pytest.raises(_hip.YpError):
        YoloPointFrontend(m, cuda, crop_resize=[0, 1, 0, 1, 2])
    # the front end changes the wrapped model while it lives (heads hook in its plans, frozen weights); close() / `with` gives it back
    assert m.model.heads_hook is True and "_frozen_version" in m.model.__dict__
    fe.close()
    assert m.model.heads_hook is False and "_frozen_version" not in m.model.__dict__
    with YoloPointFrontend(m, cuda, filter_pts=True) as fe2:
        assert m.model.heads_hook is True
        pts2, _, _ = fe2.process_img(img)
        assert pts2.shape[0] == 3
    assert m.model.heads_hook is False
    x = torch.rand(1, 3, 96, 128, device=cuda)
    assert torch.isfinite(m(x)["semi"]).all()            # the model still runs on its own, hook-free plans


@pytest.mark.parametrize("i", [0, 1, 2])
def test_point_tracker_bookkeeping(cuda, i):
    """PointTracker.update / get_tracks (HIP matcher + vectorised bookkeeping, descriptors kept on the device) against the
    reference's tracks captured in tests/golden/tracker.npz and the sequential oracle, frame by frame."""
    import os
    from helpers import tracking_sequence
    from yolopoint_amd.models.model_wrap import PointTracker
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tracker.npz"))
    D, frames, N, maxl, seed = (int(v) for v in g[f"tr{i}.cfg"])
    tr, orc = PointTracker(maxl, nn_thresh=0.7), po.PointTrackerOracle(maxl, nn_thresh=0.7)
    ids = [0] + list(range(2, maxl + 2))
    for f, (pts, desc) in enumerate(tracking_sequence(D, frames, N, seed)):
        tr.update(pts, torch.from_numpy(desc).to(cuda))
        orc.update(pts, desc)
        ref = g[f"tr{i}.f{f}.tracks"]
        assert tr.tracks.shape == ref.shape
        assert np.array_equal(tr.tracks[:, ids], ref[:, ids]) and np.array_equal(tr.tracks[:, ids], orc.tracks[:, ids])
        np.testing.assert_allclose(tr.tracks[:, 1], ref[:, 1], rtol=1e-5, atol=1e-6)          # fp32 match distances
        assert np.array_equal(tr.get_tracks(2)[:, ids], g[f"tr{i}.f{f}.long"][:, ids])
        assert np.array_equal(tr.get_matches(), g[f"tr{i}.f{f}.matches"])
        assert np.array_equal(tr.get_offsets(), orc.get_offsets())


def test_keypoint_array_from_device(cuda):
    from yolopoint_amd.frontend import to_keypoint_array
    rng = np.random.default_rng(1)
    pts = np.vstack((rng.integers(0, 1280, (2, 300)).astype(np.float64), rng.random((1, 300))))
    desc = rng.normal(size=(256, 300)).astype(np.float32)
    a = to_keypoint_array(pts, desc)
    b = to_keypoint_array(torch.from_numpy(pts).float().to(cuda), torch.from_numpy(desc).to(cuda))
    for k in a:
        assert np.array_equal(a[k], b[k]) and np.asarray(a[k]).dtype == np.asarray(b[k]).dtype, k
