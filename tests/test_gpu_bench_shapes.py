"""Parity at the shapes that are benchmarked (BASELINE.json configs[1..3]), not only at small ones: the plan-time autotuner
picks other kernel variants / tiles at these sizes, so they are checked against the oracle here.

  configs[1]  YOLOPoint-s, batch 8, 640x640: f32 compute path (north_star bar: 1e-3 of max|ref|, keypoint-cell argmax exact) and the
              f16 path that `python bench.py` times, replayed through the hipGraph as the benchmark does (bars below)
  configs[3]  YOLOPoint-l, batch 1, 1280x1280 f16; keypoint NMS of a 1280x1280 heat map with 4000 planted peaks; box NMS of
              [1, 100800, 85] with 30 000 multi-label candidates; both index selections exact
  configs[2]  all-parameter gradients in bf16 (the dtype of the training configs) against fp32 CPU autograd, with an explicit bar
"""
import numpy as np
import pytest
import torch

from helpers import make_model, rel_err, planted_heatmap, planted_predictions, argmax_mismatches
from oracle import net_oracle, postproc_oracle as po
from yolopoint_amd.utils import utils as U
from yolopoint_amd.utils.general_yolo import non_max_suppression

pytestmark = pytest.mark.gpu

# The 16-bit inference path is held to the FLOOR of 16-bit inference itself, measured in the same test: the oracle evaluated with
# PyTorch's half-precision arithmetic (fp16 storage of activations and BN-folded filters, fp32 accumulation:
# oracle.net_oracle.half_storage) differs from the fp32 oracle by floor(t) for every head tensor t; the HIP path must stay within
#   relative L2   <= 1.15 x floor    (measured: 1.00 - 1.10 x at configs[1] and configs[3])
#   max-abs error <= 2.5  x floor    (the maximum of ~1e7 rounding errors is a heavy-tailed statistic -- one decoded box of `pred` decides it:
#                                     measured 0.88 - 1.36 x with the autotuned kernel variants, 0.9 - 2.13 x over random variant mixtures,
#                                     YP_TUNE_RANDOM: other summation orders, while the L2 ratio stays within 1.00 - 1.10)
# and every keypoint cell whose argmax differs from the fp32 oracle's must be a near-tie of the REFERENCE within one 16-bit step of the
# logit scale (2^-10 max|semi|), no more such cells than 1.5 x the floor's own count + 5.
F16_L2, F16_MAX = 1.15, 2.5


def _heads(o):
    return {"semi": o["semi"], "desc": o["desc"], "pred": o["objects"][0], **{f"raw{i}": t for i, t in enumerate(o["objects"][1])}}


def _check_against_f16_floor(got, ref, floor):
    G, R, F_ = _heads(got), _heads(ref), _heads(floor)
    for k in R:
        f_max, f_l2 = rel_err(F_[k], R[k])
        g_max, g_l2 = rel_err(G[k], R[k])
        print(f"{k:5s} floor l2 {f_l2:.3e} max {f_max:.3e} | hip l2 {g_l2:.3e} ({g_l2 / f_l2:.2f}x) max {g_max:.3e} ({g_max / f_max:.2f}x)")
        assert g_l2 <= F16_L2 * f_l2, (k, g_l2, f_l2)
        assert g_max <= F16_MAX * f_max, (k, g_max, f_max)
    tie = float(ref["semi"].abs().max()) * 2.0 ** -10
    n_floor, _ = argmax_mismatches(floor["semi"], ref["semi"], tie)
    n_hip, margin = argmax_mismatches(got["semi"], ref["semi"], tie)
    print(f"argmax: {n_hip} cells differ (floor: {n_floor}), largest reference margin {margin:.2e} <= one 16-bit step {tie:.2e}")
    assert n_hip <= 1.5 * n_floor + 5, (n_hip, n_floor)


def _oracle(version, sd, x):
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    with torch.no_grad():
        return net_oracle.yolopoint_forward(sd, x, version)


def _oracle_f16_floor(version, sd, x):
    with torch.no_grad(), net_oracle.half_storage(torch.float16):
        return net_oracle.yolopoint_forward(net_oracle.fused_state_dict(sd), x, version)


def test_config1_s_bs8_640_f32(cuda):
    m, sd = make_model("s", 1234, dtype="f32")
    x = net_oracle.synth_image(8, 3, 640, 640, 1234)
    ref = _oracle("s", sd, x)
    with torch.no_grad():
        got = m.to(cuda)(x.to(cuda))
    for name in ("semi", "desc"):
        e_max, e_l2 = rel_err(got[name], ref[name])
        assert e_max < 1e-3, (name, e_max, e_l2)
    # keypoint cell argmax over 51 200 cells: identical except where the reference's own top two logits tie to fp32 rounding
    # (margin < 2e-5 at |logit| ~ 10: a different summation order decides such a cell either way)
    nbad, margin = argmax_mismatches(got["semi"], ref["semi"], 2e-5)
    print(f"argmax: {nbad} of {ref['semi'][:, 0].numel()} cells differ, all ties (largest margin {margin:.2e})")
    assert nbad <= 5
    for a, b in zip(got["objects"][1], ref["objects"][1]):         # raw Detect logits: the north-star bar
        assert rel_err(a, b)[0] < 1e-3
    # decoded rows: wh = (2 sigmoid(t))^2 * anchor amplifies a logit error by up to 2(1 - sigmoid) * wh; bar 2e-3 of max|ref|, 1e-4 in L2
    e_max, e_l2 = rel_err(got["objects"][0], ref["objects"][0])
    assert e_max < 2e-3 and e_l2 < 1e-4, (e_max, e_l2)


def test_config1_s_bs8_640_f16_graph_as_benchmarked(cuda):
    """The exact configuration `python bench.py` times: fused BN, f16, static outputs, hipGraph replay."""
    m, sd = make_model("s", 1234, dtype="f16")
    x = net_oracle.synth_image(8, 3, 640, 640, 1234)
    ref = _oracle("s", sd, x)
    m = m.to(cuda)
    m.fuse()
    m.model.use_graph = True
    with torch.no_grad():
        m(x.to(cuda))
        got = m(x.to(cuda))                        # second call = graph replay
    _check_against_f16_floor(got, ref, _oracle_f16_floor("s", sd, x))


def test_v52_s_bs8_640_f16_graph_as_benchmarked(cuda):
    """The `v52` record of bench.py: YOLOPointv52-s (the model reference configs/kitti_inference.yaml:2 selects), batch 8, 640x640, f16,
    fused BN, hipGraph replay -- against the fp32 oracle, held to the fp16 floor of the v52 network."""
    m, sd = make_model("s", 1234, dtype="f16", model_name="YOLOPointv52")
    x = net_oracle.synth_image(8, 3, 640, 640, 1234)
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    with torch.no_grad():
        ref = net_oracle.yolopointv52_forward(sd, x, "s")
        with net_oracle.half_storage(torch.float16):
            floor = net_oracle.yolopointv52_forward(net_oracle.fused_state_dict(sd), x, "s")
    m = m.to(cuda)
    m.fuse()
    m.model.use_graph = True
    with torch.no_grad():
        m(x.to(cuda))
        got = m(x.to(cuda))
    _check_against_f16_floor(got, ref, floor)


def test_config1_forward_is_deterministic(cuda):
    """Six replays of the benchmarked plan (three eager, three through the hipGraph) are bit-identical.  Regression test: the fused
    Bottleneck kernel refilled a filter-ring stage one barrier after its last read WITHOUT retiring the reads first -- the compiler had
    sunk the last MFMAs (and the lgkmcnt wait for their operands) below that barrier, so at 4 workgroups per CU a DMA occasionally landed
    under an in-flight ds_read: a handful of wrong 8x16 tiles per forward, different ones every run (only at this size)."""
    m, _ = make_model("s", 1234, dtype="f16")
    x = net_oracle.synth_image(8, 3, 640, 640, 1234).to(cuda)
    m = m.to(cuda)
    m.fuse()
    outs = []
    with torch.no_grad():
        for i in range(6):
            if i == 3:
                m.model.use_graph = True
            o = m(x)
            outs.append((o["semi"].clone(), o["desc"].clone(), o["objects"][0].clone()))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)


def test_config3_l_bs1_1280_f16(cuda):
    m, sd = make_model("l", 77, dtype="f16")
    x = net_oracle.synth_image(1, 3, 1280, 1280, 77)
    ref = _oracle("l", sd, x)
    m = m.to(cuda)
    m.fuse()
    with torch.no_grad():
        got = m(x.to(cuda))
    assert got["objects"][0].shape == (1, 100800, 85) and got["desc"].shape == (1, 256, 160, 160)
    _check_against_f16_floor(got, ref, _oracle_f16_floor("l", sd, x))


def test_config3_keypoint_nms_1280_4000_peaks(cuda):
    heat = planted_heatmap(1280, 1280, 4000, seed=4000)
    ref = po.get_pts_from_heatmap(heat, 0.015, 4)
    got = U.getPtsFromHeatmap(heat, 0.015, 4)
    assert got.shape == ref.shape and ref.shape[1] > 2000
    assert np.array_equal(got[:2], ref[:2])
    assert np.array_equal(got[2].astype(np.float32), ref[2].astype(np.float32))


def test_config3_box_nms_100800_rows_30000_candidates(cuda):
    """[1, 100800, 85] (the 1280x1280 prediction tensor): 12 000 planted rows above the confidence threshold expand to > 30 000
    multi-label candidates (max_nms = 30000 truncation of general_yolo.py:154-155,207 is exercised)."""
    pred = planted_predictions(1, 100800, 80, 12000, seed=5, img=1280)
    pred[0, :, 5:] = np.where(pred[0, :, 4:5] > 0.25, np.maximum(pred[0, :, 5:], np.random.default_rng(6).uniform(0.0, 1.0, (100800, 80)) ** 0.5), pred[0, :, 5:])
    ncand = int(((pred[0, :, 5:] * pred[0, :, 4:5] > 0.25) & (pred[0, :, 4:5] > 0.25)).sum())
    assert ncand > 30000, ncand
    for ag in (True, False):
        ref = po.non_max_suppression(pred, 0.25, 0.45, agnostic=ag, multi_label=True, max_det=300)
        got = non_max_suppression(torch.from_numpy(pred).to(cuda), 0.25, 0.45, agnostic=ag, multi_label=True, labels=[], max_det=300)
        assert got[0].shape == ref[0].shape and ref[0].shape[0] > 50
        np.testing.assert_array_equal(got[0].cpu().numpy(), ref[0])


# bf16 operands (8 mantissa bits) through ~70 layers forward and backward: against fp32 autograd, the gradients of a random-projection
# loss differ by 0.26-0.30 relative L2 in the median and 0.45-0.55 at worst -- and PyTorch's own bf16 (CPU autocast through the
# oracle) differs from fp32 by the same amounts (median 0.29, p90 0.42, worst 0.47 measured).  That noise floor is the yardstick: the
# HIP bf16 path must be no further from fp32 than PyTorch's bf16 path is (x 1.15 on median / p90, x 1.25 + 0.03 on the worst tensor),
# and every gradient must point the way of the fp32 one (cosine > 0.8).
BF16_REL = dict(median=1.15, p90=1.15, worst=1.25, worst_abs=0.03, cosine=0.8)


def _fp32_and_autocast_grads(sd, x, version, seed):
    out = {}
    for auto in (False, True):
        leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
        if auto:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
        else:
            o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
        o = {"semi": o["semi"].float(), "desc": o["desc"].float(), "objects": [t.float() for t in o["objects"]]}
        proj = net_oracle.output_projections(o, seed)
        net_oracle.projected_loss(o, proj).backward()
        out[auto] = (leaf, o, proj)
    return out


@pytest.mark.parametrize("version,B,S", [("s", 4, 128), ("n", 2, 256)])
def test_all_parameter_gradients_bf16(cuda, version, B, S):
    m, sd = make_model(version, 35, dtype="bf16")
    m = m.to(cuda).train()
    x = net_oracle.synth_image(B, 3, S, S, 35)
    ref = _fp32_and_autocast_grads(sd, x, version, 35)
    leaf, out32, proj = ref[False]
    leaf16 = ref[True][0]
    out = m(x.to(cuda))
    for k in ("semi", "desc"):
        assert rel_err(out[k], out32[k])[1] < 2.5e-2, k
    net_oracle.projected_loss(out, proj, cuda).backward()
    hip, torch16 = [], []
    for name, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        g32 = leaf[name].grad
        hip.append((rel_err(p.grad, g32)[1], name))
        torch16.append((rel_err(leaf16[name].grad, g32)[1], name))
        cos = float(torch.nn.functional.cosine_similarity(p.grad.detach().cpu().flatten().double(), g32.flatten().double(), dim=0))
        assert cos > BF16_REL["cosine"], (name, cos)
    hip.sort(); torch16.sort()
    n = len(hip)
    stats = lambda e: (e[n // 2][0], e[int(n * 0.9)][0], e[-1][0])
    (hm, h9, hw), (tm, t9, tw) = stats(hip), stats(torch16)
    print(f"bf16 gradient rel-L2 vs fp32: HIP median {hm:.3f} p90 {h9:.3f} worst {hw:.3f} | PyTorch CPU autocast median {tm:.3f} p90 {t9:.3f} worst {tw:.3f}")
    assert hm <= BF16_REL["median"] * tm and h9 <= BF16_REL["p90"] * t9 and hw <= BF16_REL["worst"] * tw + BF16_REL["worst_abs"]


def test_config2_gradient_spot_check_8x640(cuda):
    """configs[2] at ITS shape: YOLOPoint-s, 8 samples of 640 x 640 per GPU, bf16 -- the plans, tiles and BatchNorm group sizes the training
    benchmark runs (the all-parameter test above runs at 4 x 128 x 128, where the autotuner picks other kernels).  Ten parameter tensors
    spread over the network (stem, each backbone stage, PAN, the three heads) against fp32 CPU autograd through the oracle: every sampled
    gradient points the way of the fp32 one (cosine > 0.8) and stays inside the bf16 noise measured at the small shape (relative L2 <= 0.6,
    the worst small-shape tensor being 0.45-0.55); the train-mode head outputs agree to 2.5e-2."""
    version, B, S, seed = "s", 8, 640, 52
    m, sd = make_model(version, seed, dtype="bf16")
    m = m.to(cuda).train()
    x = net_oracle.synth_image(B, 3, S, S, seed)
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
    proj = net_oracle.output_projections(o, seed)
    net_oracle.projected_loss(o, proj).backward()
    out = m(x.to(cuda))
    for k in ("semi", "desc"):
        assert rel_err(out[k], o[k])[1] < 2.5e-2, k
    net_oracle.projected_loss(out, proj, cuda).backward()
    params = dict(m.named_parameters())
    names = ["model.Conv1.conv.weight", "model.Conv2.conv.weight", "model.Bottleneck1.m.0.cv2.conv.weight", "model.Conv3.bn.weight",
             "model.Bottleneck2.cv3.conv.weight", "model.Bottleneck3.m.1.cv1.conv.weight", "model.SPPooling.cv2.conv.weight",
             "model.Bottleneck6.cv1.conv.weight", "model.ConvDesc.weight", "model.Detect.m.1.weight"]
    rows = []
    for name in names:
        g, g32 = params[name].grad, leaf[name].grad
        assert g is not None and torch.isfinite(g).all(), name
        cos = float(torch.nn.functional.cosine_similarity(g.detach().cpu().flatten().double(), g32.flatten().double(), dim=0))
        rows.append((name, rel_err(g, g32)[1], cos))
    print("configs[2] shape, bf16 vs fp32 autograd: " + "; ".join(f"{n.replace('model.', '')} l2 {e:.3f} cos {c:.3f}" for n, e, c in rows))
    for name, e, c in rows:
        assert c > 0.8 and e <= 0.6, (name, e, c)


def test_config2_gradient_f32_tight_8x640(cuda):
    """configs[2] at ITS shape with a TIGHT bar: YOLOPoint-s, 8 samples of 640 x 640, the fp32 compute path (fp32 storage, exact-f32 MFMA chain),
    the plans / tiles / BatchNorm partial-row counts / weight-gradient pixel splits the autotuner builds AT THIS SHAPE -- against fp32 CPU
    autograd through the oracle at the north-star bar of 2e-3 relative L2 (the bf16 spot check above can only be judged against the bf16
    noise floor, 0.3-0.6).  Tensors whose gradient does not pass through the SPPF max pools' argmax routing (Detect, PAN, SPPF's own
    output convolution, the descriptor head): 2e-3.  Backbone tensors upstream of the pools: 2e-2 -- a max pool routes its gradient to the
    FIRST maximum of a window and two activations closer than the fp32 summation-order noise may swap (conftest.fixed_kernel_variants), which
    moves every upstream gradient by up to a percent; the kernels themselves are held to 2e-3 at the small shapes."""
    version, B, S, seed = "s", 8, 640, 53
    m, sd = make_model(version, seed, dtype="f32")
    m = m.to(cuda).train()
    x = net_oracle.synth_image(B, 3, S, S, seed)
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
    proj = net_oracle.output_projections(o, seed)
    net_oracle.projected_loss(o, proj).backward()
    out = m(x.to(cuda))
    for k in ("semi", "desc"):
        assert rel_err(out[k], o[k])[0] < 1e-3, (k, rel_err(out[k], o[k]))
    net_oracle.projected_loss(out, proj, cuda).backward()
    params = dict(m.named_parameters())
    tight = ["model.Detect.m.1.weight", "model.Bottleneck6.cv1.conv.weight", "model.ConvDesc.weight", "model.SPPooling.cv2.conv.weight", "model.ConvDet.weight"]
    upstream = ["model.Conv1.conv.weight", "model.Conv2.conv.weight", "model.Bottleneck3.m.1.cv1.conv.weight"]
    rows = []
    for name in tight + upstream:
        g, g32 = params[name].grad, leaf[name].grad
        assert g is not None and torch.isfinite(g).all(), name
        rows.append((name, rel_err(g, g32)[1]))
    print("configs[2] shape, f32 compute vs fp32 autograd: " + "; ".join(f"{n.replace('model.', '')} l2 {e:.2e}" for n, e in rows))
    for name, e in rows:
        assert e <= (2e-3 if name in tight else 2e-2), (name, e)


def test_train_bs64_batch_gradient_spot_check_128x128x128(cuda):
    """The `train_bs64` record runs ONE batch of 64 samples (128 images) per optimizer step: the same spot check at that batch -- 128
    images of 128 x 128 through one train-mode forward / backward (BatchNorm statistics over all 128 images, 2-D grids and partial-row
    counts of the large batch) against fp32 CPU autograd through the oracle."""
    version, B, S, seed = "s", 128, 128, 64
    m, sd = make_model(version, seed, dtype="bf16")
    m = m.to(cuda).train()
    x = net_oracle.synth_image(B, 3, S, S, seed)
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
    proj = net_oracle.output_projections(o, seed)
    net_oracle.projected_loss(o, proj).backward()
    out = m(x.to(cuda))
    for k in ("semi", "desc"):
        assert rel_err(out[k], o[k])[1] < 2.5e-2, k
    net_oracle.projected_loss(out, proj, cuda).backward()
    params = dict(m.named_parameters())
    names = ["model.Conv1.conv.weight", "model.Conv2.conv.weight", "model.Bottleneck1.m.0.cv2.conv.weight", "model.Conv3.bn.weight",
             "model.Bottleneck2.cv3.conv.weight", "model.Bottleneck3.m.1.cv1.conv.weight", "model.SPPooling.cv2.conv.weight",
             "model.Bottleneck6.cv1.conv.weight", "model.ConvDesc.weight", "model.Detect.m.1.weight"]
    rows = []
    for name in names:
        g, g32 = params[name].grad, leaf[name].grad
        assert g is not None and torch.isfinite(g).all(), name
        cos = float(torch.nn.functional.cosine_similarity(g.detach().cpu().flatten().double(), g32.flatten().double(), dim=0))
        rows.append((name, rel_err(g, g32)[1], cos))
    print("128 images of 128 x 128, bf16 vs fp32 autograd: " + "; ".join(f"{n.replace('model.', '')} l2 {e:.3f} cos {c:.3f}" for n, e, c in rows))
    for name, e, c in rows:
        assert c > 0.8 and e <= 0.6, (name, e, c)


def test_bf16_gradients_are_deterministic(cuda):
    """Two identical bf16 forward/backward passes give bit-identical parameter gradients (the weight-gradient reduction has a fixed
    order: per-workgroup partials folded by a second pass, no floating-point atomics)."""
    m, _ = make_model("n", 36, dtype="bf16")
    m = m.to(cuda).train()
    x = net_oracle.synth_image(2, 3, 128, 128, 36).to(cuda)
    grads = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        o = m(x)
        (o["semi"].square().mean() + o["desc"][:, :8].mean() + sum(t.tanh().mean() for t in o["objects"])).backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    names = [n for n, _ in m.named_parameters()]
    differing = [n for n, a, b in zip(names, grads[1], grads[2]) if not torch.equal(a, b)]
    assert not differing, (len(differing), differing[:12])
