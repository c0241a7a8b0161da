"""Whole-network GPU parity against the CPU oracle on seeded synthetic checkpoints.

Bars (north_star): fp32 logits / descriptors within 1e-3 relative; keypoint cell argmax bit-exact.
The f32 compute path (exact-f32 MFMA) is held to 1e-3 of max|ref| on every output; the f16 path
(fp32 accumulate, fp32 head outputs) to 1e-3 in relative L2 and 1e-2 of max|ref|."""
import pytest
import torch

from helpers import make_model, rel_err
from oracle import net_oracle

pytestmark = pytest.mark.gpu


def run_both(version, seed, B, S, dtype, cuda):
    m, sd = make_model(version, seed, dtype=dtype)
    x = net_oracle.synth_image(B, 3, S, S, seed)
    with torch.no_grad():
        ref = net_oracle.yolopoint_forward(sd, x, version)
        got = m.to(cuda)(x.to(cuda))
    return got, ref


@pytest.mark.parametrize("version,B,S", [("n", 2, 64), ("n", 1, 256), ("s", 2, 64), ("s", 1, 256), ("m", 1, 128)])
def test_forward_f32(cuda, version, B, S):
    got, ref = run_both(version, 21, B, S, "f32", cuda)
    for name in ("semi", "desc"):
        e_max, e_l2 = rel_err(got[name], ref[name])
        assert e_max < 1e-3, (name, e_max, e_l2)
    # keypoint cell argmax: bit-exact
    assert torch.equal(got["semi"].argmax(1).cpu(), ref["semi"].argmax(1))
    z, xs = got["objects"]
    zr, xr = ref["objects"]
    assert z.shape == zr.shape
    e_max, _ = rel_err(z, zr)
    assert e_max < 1e-3, ("pred", e_max)
    for a, b in zip(xs, xr):
        assert a.shape == b.shape
        assert rel_err(a, b)[0] < 1e-3


@pytest.mark.parametrize("version,B,H,W", [("n", 3, 96, 160), ("s", 5, 64, 224), ("n", 1, 32, 32), ("s", 7, 160, 96), ("n", 9, 64, 64)])
def test_forward_non_square_and_odd_batches(cuda, version, B, H, W):
    """Ragged shapes: non-square inputs (multiples of 32), batch sizes that are not powers of two, the minimum 32 x 32 input."""
    for dtype, bar in (("f32", 1e-4), ("f16", 3e-3)):
        m, sd = make_model(version, 31, dtype=dtype)
        x = net_oracle.synth_image(B, 3, H, W, 5)
        with torch.no_grad():
            ref = net_oracle.yolopoint_forward(sd, x, version)
            got = m.to(cuda)(x.to(cuda))
        for name in ("semi", "desc"):
            assert got[name].shape == ref[name].shape
            assert rel_err(got[name], ref[name])[1] < bar, (dtype, name)
        assert got["objects"][0].shape == ref["objects"][0].shape
        assert rel_err(got["objects"][0], ref["objects"][0])[1] < (1e-4 if dtype == "f32" else 2e-2)
        if dtype == "f32":
            assert torch.equal(got["semi"].argmax(1).cpu(), ref["semi"].argmax(1))


# 16-bit perf paths: measured error grows with depth (fp16 rounding of weights and of every
# layer's activations, fp32 accumulation): raw head logits are the quantities held to a bar; the
# decoded boxes ((2*sigmoid)^2 * anchor amplifies logit error) get a looser L2 bar.
TOL16 = {"f16": dict(head_l2=3e-3, head_max=1e-2, raw_l2=1e-2, pred_l2=2e-2),
         "bf16": dict(head_l2=2.5e-2, head_max=8e-2, raw_l2=6e-2, pred_l2=1e-1)}


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("version,B,S", [("n", 2, 64), ("s", 1, 256)])
def test_forward_16bit(cuda, version, B, S, dtype):
    got, ref = run_both(version, 22, B, S, dtype, cuda)
    t = TOL16[dtype]
    errs = {name: rel_err(got[name], ref[name]) for name in ("semi", "desc")}
    for i, (a, b) in enumerate(zip(got["objects"][1], ref["objects"][1])):
        errs[f"x{i}"] = rel_err(a, b)
    errs["pred"] = rel_err(got["objects"][0], ref["objects"][0])
    print(dtype, version, {k: (f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in errs.items()})
    for name in ("semi", "desc"):
        assert errs[name][1] < t["head_l2"] and errs[name][0] < t["head_max"], (dtype, errs)
    for i in range(3):
        assert errs[f"x{i}"][1] < t["raw_l2"], (dtype, errs)
    assert errs["pred"][1] < t["pred_l2"], (dtype, errs)
    # keypoint cell argmax may differ only where the top-2 logits are within the 16-bit error
    same = (got["semi"].argmax(1).cpu() == ref["semi"].argmax(1)).float().mean()
    assert same > 0.97, float(same)


def test_fuse_matches_unfused(cuda):
    """Model.fuse() (reference YOLOPoint.py:84-90) must not change the outputs; state_dict shrinks to conv.{weight,bias}."""
    m, sd = make_model("n", 5, dtype="f32")
    x = net_oracle.synth_image(1, 3, 64, 64, 5).to(cuda)
    m = m.to(cuda)
    a = m(x)
    m.fuse()
    keys = list(m.state_dict().keys())
    assert "model.Conv1.conv.bias" in keys and not any(".bn." in k for k in keys)
    b = m(x)
    assert rel_err(b["semi"], a["semi"])[0] < 1e-5 and rel_err(b["desc"], a["desc"])[0] < 1e-5


def test_frozen_weights_skip_the_version_walk_but_not_invalidations(cuda):
    """HipModule.freeze_weights() (what frontend.YoloPointFrontend sets): no per-forward walk over the parameters' version counters; an
    in-place edit of a parameter is then not seen -- the documented contract -- while invalidate_packed_weights() / an optimizer step /
    load_state_dict (the generation counter) still re-derive the packed filters, and freeze_weights(False) restores the per-forward check."""
    from yolopoint_amd.models.common import invalidate_packed_weights
    m, sd = make_model("n", 3, dtype="f16")
    m = m.to(cuda)
    x = net_oracle.synth_image(1, 3, 64, 64, 3).to(cuda)
    net = m.model
    a = m(x)["semi"].clone()
    net.freeze_weights()
    with torch.no_grad():
        net.ConvDet.weight.mul_(2.0)
    assert torch.equal(m(x)["semi"], a)                 # frozen: the edit is not seen
    invalidate_packed_weights()
    b = m(x)["semi"].clone()
    assert not torch.equal(b, a) and rel_err(b, 2.0 * a)[0] < 2e-3
    net.freeze_weights(False)
    with torch.no_grad():
        net.ConvDet.weight.mul_(0.5)
    assert rel_err(m(x)["semi"], a)[0] < 2e-3           # unfrozen: version counters are read again


def test_cpu_tensor_raises():
    from yolopoint_amd import _hip
    m, _ = make_model("n", 1)
    with pytest.raises(_hip.YpError):
        m(torch.zeros(1, 3, 64, 64))


def test_multistream_graph_equals_eager(cuda, monkeypatch):
    """The one-lane hipGraph replay (YP_INFER_LANES=0: branches as graph edges) and the two-lane replay must reproduce the eager,
    single-stream launch order bit for bit -- a schedule only reorders independent launches."""
    monkeypatch.setenv("YP_INFER_LANES", "0")
    m, _ = make_model("s", 9, dtype="f16")
    m = m.to(cuda)
    x = net_oracle.synth_image(2, 3, 128, 128, 9).to(cuda)
    a = m(x)
    m.model.use_graph = True
    b = m(x)
    c = m(x)      # replay
    plan = next(iter(m.model._plans.values()))[0]
    assert plan.graph and plan.parallel
    monkeypatch.setenv("YP_INFER_LANES", "1")
    m.model._plans.clear()
    d = m(x)
    plan = next(iter(m.model._plans.values()))[0]
    assert plan.has_lanes and plan.parallel
    for k in ("semi", "desc"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]) and torch.equal(a[k], d[k]), k
    assert torch.equal(a["objects"][0], b["objects"][0]) and torch.equal(a["objects"][0], c["objects"][0]) and torch.equal(a["objects"][0], d["objects"][0])


@pytest.mark.parametrize("capture", [False, True])
@pytest.mark.parametrize("model_name", ["YOLOPoint", "YOLOPointv52"])
def test_two_lane_schedule_tracks_changing_inputs(cuda, monkeypatch, capture, model_name):
    """The inference plan runs the keypoint / descriptor heads and two Detect levels on a side lane (PlanBuilder.side).  Every replay must
    equal the one-lane result for ITS OWN input, bit for bit: with a static input a missing dependency hides behind the previous replay's
    (identical) values -- which is how a captured graph whose main chain had lost an edge (every side op re-forking from the same node:
    ROCm 7.2 then started the next main-lane kernel early) passed every fixed-input test while it was a third faster than legal.
    capture=True: the lanes as a forked branch of a hipGraph (YP_LANES_EAGER=0); False: two plain streams (the default)."""
    m, _ = make_model("s", 21, dtype="f16", model_name=model_name)
    m = m.to(cuda)
    m.fuse()
    xs = [net_oracle.synth_image(4, 3, 256, 256, 300 + i).to(cuda) for i in range(3)]

    def grab(o):
        return [o["semi"].clone(), o["desc"].clone(), o["objects"][0].clone()] + [t.clone() for t in o["objects"][1]]
    with torch.no_grad():
        monkeypatch.setenv("YP_INFER_LANES", "0")
        ref = [grab(m(x)) for x in xs]
        monkeypatch.setenv("YP_INFER_LANES", "1")
        monkeypatch.setenv("YP_LANES_EAGER", "0" if capture else "1")
        m.model._plans.clear()
        m.model.use_graph = True
        for rnd in range(2):
            for i, x in enumerate(xs):
                got = grab(m(x))
                plan = next(iter(m.model._plans.values()))[0]
                assert plan.has_lanes and plan.graph == capture
                for j, (g, r) in enumerate(zip(got, ref[i])):
                    assert torch.equal(g, r), (rnd, i, j)


@pytest.mark.parametrize("version,B,S,dtype", [("n", 2, 64, "f32"), ("s", 1, 128, "f32"), ("s", 1, 128, "f16")])
def test_forward_v52(cuda, version, B, S, dtype):
    """YOLOPointv52 (C2f blocks, MaxPool2d descriptor branch, 65-channel C2f keypoint head) against the oracle."""
    m, sd = make_model(version, 23, dtype=dtype, model_name="YOLOPointv52")
    x = net_oracle.synth_image(B, 3, S, S, 23)
    with torch.no_grad():
        ref = net_oracle.yolopointv52_forward(sd, x, version)
        got = m.to(cuda)(x.to(cuda))
    tol = 1e-3 if dtype == "f32" else 1e-2
    for name in ("semi", "desc"):
        assert got[name].shape == ref[name].shape
        assert rel_err(got[name], ref[name])[0] < tol, (name, rel_err(got[name], ref[name]))
    assert rel_err(got["objects"][0], ref["objects"][0])[1 if dtype != "f32" else 0] < (1e-3 if dtype == "f32" else 2e-2)
    if dtype == "f32":
        assert torch.equal(got["semi"].argmax(1).cpu(), ref["semi"].argmax(1))


def test_stream_pick_returns_tested_companions(cuda):
    """yp_stream_pick: the companion streams of a caller's stream come from the library's pool, differ from the caller's and from each
    other, stay the same for the life of the process, and -- the point of the test kernels -- work queued on slot 0 completes while the
    caller's stream is still busy (different hardware queues)."""
    import ctypes as C
    from yolopoint_amd import _hip
    l = _hip.lib()
    main = torch.cuda.Stream(device=cuda)
    picks = []
    for slot in (0, 1, 0, 1):
        out = C.c_void_p()
        _hip.check(l.yp_stream_pick(C.c_void_p(main.cuda_stream), slot, C.byref(out)))
        picks.append(out.value)
    assert picks[0] == picks[2] and picks[1] == picks[3]
    assert len({main.cuda_stream, picks[0], picks[1]}) == 3
    side = torch.cuda.ExternalStream(picks[0], device=cuda)
    x = torch.randn(4096, 4096, device=cuda)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        for _ in range(40):                       # ~ms of work on the caller's stream
            x = x @ x * 1e-3
        busy = main.record_event()
    with torch.cuda.stream(side):
        y = torch.zeros(8, device=cuda) + 1
        done = side.record_event()
    done.synchronize()
    assert not busy.query(), "the side lane's work waited for the caller's stream: the two share a hardware queue"
    torch.cuda.synchronize()
    assert float(y.sum()) == 8.0
    # a caller that destroys its stream tells the library: the picks are made (and tested) again for whatever stream shows up at that address
    _hip.check(l.yp_stream_forget(C.c_void_p(main.cuda_stream)))
    out = C.c_void_p()
    _hip.check(l.yp_stream_pick(C.c_void_p(main.cuda_stream), 0, C.byref(out)))
    assert out.value is not None and out.value != main.cuda_stream
    _hip.check(l.yp_stream_forget(C.c_void_p(-1)))


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("model_name", ["YOLOPoint", "YOLOPointv52"])
def test_fused_stem_conv2_equals_the_two_launches(cuda, monkeypatch, dtype, model_name):
    """Conv1 + Conv2 as one launch (stem output in LDS only; -s width) against the two-launch plan: non-square input whose Conv2 tiles are
    ragged in both directions (H / 4 = 40 rows = 10 tiles of 4, W / 4 = 56 columns = 3.5 tiles of 16), image borders on every side.  The
    stem output is rounded to 16 bits in both forms and Conv2 sums its taps in the same order: every head must agree to the last bit
    or, where the two-launch plan's tile sums in another order, to 2 ulp of the 16-bit storage."""
    from helpers import make_model
    from oracle import net_oracle
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("YP_FUSE_STEM2", fuse)
        m, _ = make_model("s", 29, dtype=dtype, model_name=model_name)
        m = m.to(cuda).eval()
        m.fuse()
        x = net_oracle.synth_image(2, 3, 160, 224, 31).to(cuda)
        with torch.no_grad():
            o = m(x)
        plan = next(iter(m.model._plans.values()))[0]
        assert ("Conv2" in plan.stem_record.name) == (fuse == "1"), plan.stem_record.name
        outs[fuse] = [o["semi"].float().clone(), o["desc"].float().clone(), o["objects"][0].float().clone()]
    for a, b in zip(outs["1"], outs["0"]):
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()) * (1 if dtype == "bf16" else 0.125) + 1e-6


def test_fused_stem_equivalence_needs_order_preserving_variants_on_the_unfused_layers(cuda, monkeypatch):
    """Root cause of round 5's "the wave-private split-K tiles (71..76) broke the fused-stem equivalence tests under a random variant
    mixture" (tools/probe/wsk_rootcause.py, profiles/r06_wsk_rootcause.txt: 34 arms).  The equivalence tests above compare a fused-stem plan with
    a two-launch plan; every layer the two plans SHARE draws the same variant in both (the tuner caches per signature), so only the layers that
    exist in ONE of the plans -- `Conv2`, `Bottleneck1.cv1+cv2` of the two-launch form -- can make them differ, and they do so exactly when
    their variant sums k in another order than the fused stem's phases do.  Every failing mixture had drawn a split-K tile for one of those
    two layers (4 of 4; the 4 mixtures with split-K tiles on shared layers only agreed bit for bit), and the size of the disagreement is what
    ANY change of summation order produces at the decoded Detect head (two order-preserving mixtures against each other: 7-84 x the test's
    tolerance; the failing ones 10-68 x).  No buffer was corrupted.  Restated here with the product's own order-changing tiles (31 / 33,
    waves-split-k, reachable by explicit id only for this reason):
      * default variants: the fused plan and the two-launch plan agree to the equivalence tests' tolerance;
      * the two-launch plan with Conv2 / Bottleneck1.cv1+cv2 forced onto a split-K tile: the heads DIFFER from the fused plan's (the
        forced tile took effect and reorders the fp32 sums) -- and that plan is exactly as far from the fp32 oracle as the fused one
        (relative L2 of semi / desc within 25 % of each other, the heavy-tailed decoded Detect rows within the same order of magnitude, 2-3e-3): rounding noise, not a damaged buffer."""
    from helpers import make_model, rel_err
    from oracle import net_oracle
    from yolopoint_amd.plan import PlanBuilder
    forced = []

    def tuner(self, d, det, key, stat_group_px=None):
        name = self.name()
        if os.environ.get("YP_TEST_SPLITK") == "1" and (name.startswith("Conv2") or name.startswith("Bottleneck1.cv1")):
            for tile in (31, 33):
                d.tile = tile
                if _hip.lib().yp_conv2d(C.byref(d), _hip.stream_ptr()) == 0:
                    forced.append((name, tile))
                    return tile, None
        return 0, None
    import ctypes as C
    import os
    from yolopoint_amd import _hip
    monkeypatch.setattr(PlanBuilder, "_autotune", tuner)
    x = net_oracle.synth_image(2, 3, 160, 224, 31)
    outs, ref = {}, None
    for arm, fuse, splitk in (("fused", "1", "0"), ("two", "0", "0"), ("two_splitk", "0", "1")):
        monkeypatch.setenv("YP_FUSE_STEM2", fuse)
        monkeypatch.setenv("YP_TEST_SPLITK", splitk)
        m, sd = make_model("s", 29, dtype="f16")
        if ref is None:
            with torch.no_grad():
                r = net_oracle.yolopoint_forward(sd, x, "s")
            ref = [r["semi"], r["desc"], r["objects"][0]]
        m = m.to(cuda).eval()
        m.fuse()
        with torch.no_grad():
            o = m(x.to(cuda))
        outs[arm] = [o["semi"].float().cpu(), o["desc"].float().cpu(), o["objects"][0].float().cpu()]
    tol = lambda b: 2.0 ** -7 * float(b.abs().max()) * 0.125 + 1e-6
    for a, b in zip(outs["fused"], outs["two"]):
        assert float((a - b).abs().max()) <= tol(b)
    assert {n.split(".")[0] for n, _ in forced} == {"Conv2", "Bottleneck1"}, forced
    differing = [float((a != b).float().mean()) for a, b in zip(outs["fused"], outs["two_splitk"])]
    assert min(differing) > 0.01, differing                 # another summation order in two early layers: every head moves
    for k, (a, b, r_) in enumerate(zip(outs["fused"], outs["two_splitk"], ref)):
        ea, eb = rel_err(a, r_)[1], rel_err(b, r_)[1]
        print(f"head {k}: fused vs oracle {ea:.3e}, two-launch with split-K tiles vs oracle {eb:.3e}, differing elements {differing[k]:.3f}")
        if k < 2:
            assert abs(ea - eb) <= 0.25 * max(ea, eb), (k, ea, eb)          # (a damaged buffer is O(1): 100 x these errors)
        else:       # decoded Detect rows: (2 sigmoid)^2 x anchor makes the L2 error heavy-tailed (a handful of rows carry it): same order of magnitude
            assert max(ea, eb) <= 2e-2 and max(ea, eb) <= 4.0 * min(ea, eb), (k, ea, eb)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("model_name", ["YOLOPoint", "YOLOPointv52"])
def test_fused_stem_conv2_c3_head_equals_the_separate_launch(cuda, monkeypatch, dtype, model_name):
    """Conv1 + Conv2 + Bottleneck1.cv1/cv2 as ONE launch (Conv2's output lives in LDS only; YP_FUSE_STEM3) against the plan with the
    pointwise launch behind the fused stem + Conv2: the same ragged, non-square input; Conv2's output is rounded to 16 bits in both forms
    and the pointwise filter sums its 64 inputs in two 32-deep steps in both: heads to 2 ulp of the 16-bit storage.  Also at batch 1
    and with an input that changes between replays (the fused launch sits outside the plan: its destinations are plan buffers)."""
    from helpers import make_model
    from oracle import net_oracle
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("YP_FUSE_STEM3", fuse)
        m, _ = make_model("s", 29, dtype=dtype, model_name=model_name)
        m = m.to(cuda).eval()
        m.fuse()
        res = []
        for seed, B in ((31, 2), (32, 2), (33, 1)):
            x = net_oracle.synth_image(B, 3, 160, 224, seed).to(cuda)
            with torch.no_grad():
                o = m(x)
            plan = next(iter(m.model._plans.values()))[0]
            assert ("Bottleneck1.cv1" in plan.stem_record.name) == (fuse == "1"), plan.stem_record.name
            res.append([o["semi"].float().clone(), o["desc"].float().clone(), o["objects"][0].float().clone()])
        outs[fuse] = res
    for ra, rb in zip(outs["1"], outs["0"]):
        for a, b in zip(ra, rb):
            assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()) * (1 if dtype == "bf16" else 0.125) + 1e-6
