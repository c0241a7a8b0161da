"""CPU: the oracle restatement reproduces the golden outputs captured from the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import net_oracle, postproc_oracle as po

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lay():
    with open(os.path.join(G, "layouts.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def net():
    return np.load(os.path.join(G, "network.npz"))


@pytest.fixture(scope="module")
def post():
    return np.load(os.path.join(G, "postproc.npz"))


@pytest.mark.parametrize("tag,v,B,S,seed", [("n64", "n", 2, 64, 21), ("n128", "n", 1, 128, 21), ("s64", "s", 2, 64, 21),
                                            ("n64b", "n", 2, 64, 22), ("s128", "s", 1, 128, 22)])
def test_network_eval(lay, net, tag, v, B, S, seed):
    layout = [(k, tuple(s)) for k, s in lay[v]["state_dict"]]
    sd = net_oracle.synth_state_dict(layout, seed)
    x = net_oracle.synth_image(B, 3, S, S, seed)
    with torch.no_grad():
        o = net_oracle.yolopoint_forward(sd, x, v)
    np.testing.assert_allclose(o["semi"].numpy(), net[f"{tag}.semi"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["desc"].numpy(), net[f"{tag}.desc"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["objects"][0].numpy(), net[f"{tag}.pred"], rtol=1e-4, atol=1e-4)
    for i, t in enumerate(o["objects"][1]):
        np.testing.assert_allclose(t.numpy(), net[f"{tag}.x{i}"], rtol=1e-4, atol=1e-5)
    # keypoint cell argmax: exact
    assert np.array_equal(o["semi"].argmax(1).numpy(), net[f"{tag}.semi"].argmax(1))


@pytest.mark.parametrize("tag,v,B,S,seed", [("v52n64", "n", 2, 64, 23), ("v52s128", "s", 1, 128, 23)])
def test_network_v52(lay, net, tag, v, B, S, seed):
    layout = [(k, tuple(s)) for k, s in lay["v52_" + v]["state_dict"]]
    sd = net_oracle.synth_state_dict(layout, seed)
    x = net_oracle.synth_image(B, 3, S, S, seed)
    with torch.no_grad():
        o = net_oracle.yolopointv52_forward(sd, x, v)
    np.testing.assert_allclose(o["semi"].numpy(), net[f"{tag}.semi"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["desc"].numpy(), net[f"{tag}.desc"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["objects"][0].numpy(), net[f"{tag}.pred"], rtol=1e-4, atol=1e-4)


def test_network_train_mode_and_fuse(lay, net):
    layout = [(k, tuple(s)) for k, s in lay["n"]["state_dict"]]
    sd = net_oracle.synth_state_dict(layout, 21)
    x = net_oracle.synth_image(2, 3, 64, 64, 21)
    stats = {}
    with torch.no_grad():
        o = net_oracle.yolopoint_forward(sd, x, "n", training=True, stats=stats)
    np.testing.assert_allclose(o["semi"].numpy(), net["n64.train.semi"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["desc"].numpy(), net["n64.train.desc"], rtol=1e-4, atol=1e-5)
    for i, t in enumerate(o["objects"]):
        np.testing.assert_allclose(t.numpy(), net[f"n64.train.x{i}"], rtol=1e-4, atol=1e-5)
    for k in ("model.Conv1.bn.running_mean", "model.Conv1.bn.running_var", "model.Bottleneck8.cv3.bn.running_mean",
              "model.Bottleneck8.cv3.bn.running_var"):
        np.testing.assert_allclose(stats[k].numpy(), net["n64.train." + k], rtol=1e-5, atol=1e-6)
    # folding BN by the oracle's formula == the reference's fused model
    fused = dict(sd)
    for k in [k for k in sd if k.endswith(".bn.weight")]:
        p = k[:-len(".bn.weight")]
        w, b = net_oracle.fuse_conv_bn(sd[p + ".conv.weight"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                                       sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"])
        fused[p + ".conv.weight"], fused[p + ".conv.bias"] = w, b
        for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            del fused[f"{p}.bn.{s}"]
    with torch.no_grad():
        of = net_oracle.yolopoint_forward(fused, x, "n")
    np.testing.assert_allclose(of["semi"].numpy(), net["n64.fused.semi"], rtol=1e-4, atol=2e-4)   # |semi| up to ~40: fp32 re-association noise


def test_blocks():
    blk = np.load(os.path.join(G, "blocks.npz"))
    from yolopoint_amd.models.common import Conv, Bottleneck, C3, SPPF     # only used for the parameter layout
    cases = [("conv_stem", Conv(3, 16, 6, 2, 2), 3, 2, 32, 32, lambda sd, x: net_oracle.conv_block(sd, "blk", x, 6, 2, 2)),
             ("conv_3x3s2", Conv(16, 32, 3, 2), 16, 2, 16, 16, lambda sd, x: net_oracle.conv_block(sd, "blk", x, 3, 2, 1)),
             ("conv_1x1", Conv(32, 24, 1, 1), 32, 1, 9, 7, lambda sd, x: net_oracle.conv_block(sd, "blk", x, 1, 1, 0)),
             ("bottleneck", Bottleneck(32, 32, True, e=1.0), 32, 2, 10, 10, lambda sd, x: net_oracle.bottleneck(sd, "blk", x)),
             ("c3_n2", C3(64, 32, 2), 64, 1, 12, 12, lambda sd, x: net_oracle.c3(sd, "blk", x, 2)),
             ("sppf", SPPF(64, 64, 5), 64, 2, 9, 11, lambda sd, x: net_oracle.sppf(sd, "blk", x))]
    for tag, mod, c1, B, H, W, fn in cases:
        layout = [("blk." + k, tuple(v.shape)) for k, v in mod.state_dict().items()]
        sd = net_oracle.synth_state_dict(layout, 11)
        x = net_oracle.synth_image(B, c1, H, W, 5) - 0.5
        np.testing.assert_allclose(fn(sd, x).numpy(), blk[tag], rtol=1e-4, atol=1e-5, err_msg=tag)


def test_flatten_detection(post):
    np.testing.assert_allclose(po.flatten_detection(post["flat.semi4"]), post["flat.out4"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(po.flatten_detection(post["flat.semi3"]), post["flat.out3"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(po.flatten_detection_demo(post["flat.semi3"]), post["flat.demo3"], rtol=1e-5, atol=1e-8)


def test_keypoint_nms(post):
    i = 0
    while f"kp{i}.cfg" in post:
        H, W, npk, thr, r = post[f"kp{i}.cfg"]
        H, W, npk, r = int(H), int(W), int(npk), int(r)
        heat = helpers.planted_heatmap(H, W, npk, seed=H + npk + r)
        if npk == 0:
            heat = (heat * 0.0 + 0.01).astype(np.float32)
        got = po.get_pts_from_heatmap(heat, thr, r)
        assert got.shape == post[f"kp{i}.pts"].shape and np.array_equal(got, post[f"kp{i}.pts"]), i
        i += 1
    assert i == 5
    for name, r in (("edge1", 4), ("edge2", 4), ("edge3", 2)):
        assert np.array_equal(po.get_pts_from_heatmap(post[f"kp.{name}.heat"], 0.1, r), post[f"kp.{name}.pts"]), name
    o, idx = po.nms_fast(post["nmsfast.in"], 60, 80, 4)
    assert np.array_equal(o, post["nmsfast.out"]) and np.array_equal(idx, post["nmsfast.idx"])
    for k in (0, 1):                       # 0- and 1-corner special cases return int arrays (utils.py:151-155)
        o, idx = po.nms_fast(post["nmsfast.in"][:, :k], 60, 80, 4)
        assert o.shape == (3, k) and idx.shape == (k,)
    assert np.array_equal(po.get_pts_from_semi(post["semi2pts.semi"], 0.05, 4)[:2], post["semi2pts.pts"][:2])


def test_box_nms(post):
    i = 0
    while f"box{i}.cfg" in post:
        B, N, nc, ncand, ml, ag, md = [int(v) for v in post[f"box{i}.cfg"]]
        pred = helpers.planted_predictions(B, N, nc, ncand, seed=N + nc + ncand)
        if ncand == 0:
            pred[..., 4] = 0.1
        dets = po.non_max_suppression(pred, 0.25, 0.45, agnostic=bool(ag), multi_label=bool(ml), max_det=md)
        for b in range(B):
            ref = post[f"box{i}.det{b}"]
            assert dets[b].shape == ref.shape and np.array_equal(dets[b], ref), (i, b)
        i += 1
    assert i == 6


def test_descriptors(post):
    np.testing.assert_allclose(po.sample_desc_from_points(post["samp.desc"], post["samp.pts"]), post["samp.out"], rtol=1e-5, atol=1e-6)
    assert po.sample_desc_from_points(post["samp.desc"], np.zeros((3, 0))).shape == (64, 0)
    i = 0
    while f"mnn{i}.cfg" in post:
        D, N1, N2 = [int(v) for v in post[f"mnn{i}.cfg"]]
        d1, d2 = helpers.planted_descriptors(D, N1, N2, 0.7, seed=N1 + N2)
        m = po.nn_match_two_way(d1, d2, 0.7)
        assert np.array_equal(m[:2], post[f"mnn{i}.matches"][:2])
        np.testing.assert_allclose(m[2], post[f"mnn{i}.matches"][2], atol=1e-6)
        i += 1
    assert i == 3
    assert po.nn_match_two_way(np.zeros((8, 0)), np.zeros((8, 3)), 0.7).shape == (3, 0)


def test_labels(post):
    np.testing.assert_allclose(po.labels2d_to_3d(post["l2d.labels"]), post["l2d.out"])
    np.testing.assert_allclose(po.get_masks(post["l2d.mask"]), post["l2d.maskout"])


# ------------------------------------------------------------------ homography adaptation (export)
@pytest.mark.parametrize("i", [0, 1, 2])
def test_homography_adaptation_oracle(i):
    """oracle/postproc_oracle.homography_adaptation against the imported reference's export flow (export_homography.py:88-150)."""
    g = np.load(os.path.join(G, "export.npz"))
    N, Hc, Wc, thr, r, top_k = g[f"ha{i}.cfg"]
    N, Hc, Wc, r, top_k = int(N), int(Hc), int(Wc), int(r), int(top_k)
    mask = g[f"ha{i}.valid_mask"].astype(np.float32)
    # the valid masks themselves: compute_valid_mask (utils/utils.py:297-331) = nearest warp of ones + a 1-px frame
    m2 = po.warp_image_batch(np.ones((N, 1, Hc * 8, Wc * 8), np.float32), g[f"ha{i}.homographies"], mode="nearest")
    m2[:, :, :1, :] = 0; m2[:, :, -1:, :] = 0; m2[:, :, :, :1] = 0; m2[:, :, :, -1:] = 0
    assert np.array_equal(m2, mask)
    agg, pts = po.homography_adaptation(g[f"ha{i}.semi"], mask, g[f"ha{i}.inv_homographies"], thr, r, top_k)
    ref = g[f"ha{i}.agg"]
    assert np.array_equal(np.isnan(agg), np.isnan(ref))
    ok = ~np.isnan(ref)
    # fp32 source coordinates carry ~1e-5 px of rounding x a heat gradient of O(0.5 / px): absolute 1e-5 on a [0,1] map
    np.testing.assert_allclose(agg[ok], ref[ok], rtol=1e-5, atol=1e-5)
    rp = g[f"ha{i}.pts"]
    assert pts.shape == rp.shape
    assert np.array_equal(pts[:, :2], rp[:, :2])
    np.testing.assert_allclose(pts[:, 2], rp[:, 2], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ PointTracker bookkeeping
@pytest.mark.parametrize("i", [0, 1, 2])
def test_point_tracker_oracle(i):
    """oracle PointTrackerOracle against the imported reference tracker (models/model_wrap.py:410-606) frame by frame."""
    g = np.load(os.path.join(G, "tracker.npz"))
    D, frames, N, maxl, seed = (int(v) for v in g[f"tr{i}.cfg"])
    tr = po.PointTrackerOracle(maxl, nn_thresh=0.7)
    for f, (pts, desc) in enumerate(helpers.tracking_sequence(D, frames, N, seed)):
        tr.update(pts, desc)
        ref = g[f"tr{i}.f{f}.tracks"]
        ids = [0] + list(range(2, maxl + 2))
        assert np.array_equal(tr.tracks[:, ids], ref[:, ids])
        np.testing.assert_allclose(tr.tracks[:, 1], ref[:, 1], rtol=1e-9)
        assert np.array_equal(tr.get_tracks(2)[:, ids], g[f"tr{i}.f{f}.long"][:, ids])
        assert np.array_equal(tr.matches, g[f"tr{i}.f{f}.matches"])


def test_keypoint_array_wire_format():
    """KeypointArray.msg fields as yolopoint_ros.py:109-117 fills them (host path of the packer)."""
    from yolopoint_amd.frontend import to_keypoint_array
    rng = np.random.default_rng(0)
    pts = np.vstack((rng.integers(0, 640, (2, 17)).astype(np.float64), rng.random((1, 17))))
    desc = rng.normal(size=(128, 17)).astype(np.float32)
    m = to_keypoint_array(pts, desc)
    assert m["x"].dtype == np.uint16 and m["y"].dtype == np.uint16 and m["score"].dtype == np.float32 and m["desc_flat"].dtype == np.float32
    assert np.array_equal(m["x"], pts[1].astype(np.uint16)) and np.array_equal(m["y"], pts[0].astype(np.uint16))
    assert np.array_equal(m["score"], pts[2].astype(np.float32))
    assert m["desc_len"] == 128 and m["desc_len"].dtype == np.uint8
    assert np.array_equal(m["desc_flat"], desc.flatten())
    assert to_keypoint_array(pts, np.zeros((256, 17), np.float32))["desc_len"] == 0


def test_box_nms_classes_and_labels():
    """non_max_suppression(classes=..., labels=...) (general_yolo.py:171-178,199-200): oracle vs the imported reference's kept rows."""
    P = np.load(os.path.join(os.path.dirname(__file__), "golden", "postproc.npz"))
    from helpers import planted_predictions
    pred = planted_predictions(2, 1000, 80, 120, seed=77)
    labels = [P["boxc.labels0"], np.zeros((0, 5), np.float32)]
    for ml in (0, 1):
        mine = po.non_max_suppression(pred, 0.25, 0.45, agnostic=False, multi_label=bool(ml), max_det=300, classes=list(P["boxc.classes"]), labels=labels)
        for b, mm in enumerate(mine):
            assert np.array_equal(mm, P[f"boxc.ml{ml}.det{b}"]), (ml, b)
