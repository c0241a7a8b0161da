"""Every registered `path` switch of yolopoint_amd/switches.py computes the same thing as the default.

The table is the registry itself: a switch that selects an alternative code path cannot exist without appearing here (sw() refuses unregistered
names, tests/test_host_layout.py::test_every_environment_switch_is_registered checks the sources for stray reads).  One small forward
(YOLOPoint-s, 2 x 128 x 160, f16, BN folded) or optimizer-step gradient (YOLOPoint-n / -s, 2 x 128 x 128, bf16 | fp8) per (switch, value), compared
with the default run of the same process at the registered tolerance: 0 = bit-identical (schedules, lanes, graph forms), otherwise the relative L2
of every head / the median relative L2 over all parameter gradients (paths that sum in another order or round elsewhere)."""
import pytest
import torch

from helpers import make_model, rel_err
from oracle import net_oracle
from yolopoint_amd.switches import SWITCHES

pytestmark = pytest.mark.gpu

CASES = [(n, v) for n, s in SWITCHES.items() if s.kind == "path" and s.scope for v in s.alt]
_cache = {}


def _infer(cuda):
    m, _ = make_model("s", 41, dtype="f16")
    m = m.to(cuda).eval()
    m.fuse()
    x = net_oracle.synth_image(2, 3, 128, 160, 41).to(cuda)
    with torch.no_grad():
        o = m(x)
    return [o["semi"].float().clone(), o["desc"].float().clone(), o["objects"][0].float().clone()]


def _train(cuda, version="n", fp8=False, steps=0, native_stage=True):
    """Gradients of one micro-batch (steps = 0) or the parameters after `steps` optimizer steps."""
    from yolopoint_amd.engine import TrainStep, synthetic_batch
    m, _ = make_model(version, 43, dtype="bf16")
    m = m.to(cuda).train()
    step = TrainStep(m, cuda, img_size=128, fp8=fp8)
    step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
    batch = synthetic_batch(2, 128, cuda, 4300)
    torch.manual_seed(77)
    if steps:
        for _ in range(steps):
            loss = step(batch)
        torch.cuda.synchronize()
        vals = [p.detach().float().clone() for p in m.parameters()]
    else:
        if fp8:                      # (the first call calibrates the delayed scales and takes one optimizer step)
            step(batch)
            torch.manual_seed(78)
        loss = step.loss_and_grads(batch)
        torch.cuda.synchronize()
        vals = [p.grad.detach().float().clone() for p in m.parameters()]
    assert all(torch.isfinite(v).all() for v in vals)
    return vals, float(loss)


def _default(key, fn):
    if key not in _cache:
        _cache[key] = fn()
    return _cache[key]


def _compare(got, ref, tol, what):
    if tol == 0.0:
        bad = [i for i, (a, b) in enumerate(zip(got, ref)) if not torch.equal(a, b)]
        assert not bad, f"{what}: {len(bad)} of {len(ref)} tensors differ from the default (first: {bad[:5]}; max |d| "\
                        f"{max(float((got[i] - ref[i]).abs().max()) for i in bad):.3e})"
        return
    errs = sorted(rel_err(a, b)[1] for a, b in zip(got, ref))
    med = errs[len(errs) // 2]
    assert med <= tol and errs[-1] <= 50 * tol, f"{what}: relative L2 median {med:.3e} worst {errs[-1]:.3e} (bar {tol:.1e})"


@pytest.mark.parametrize("name,value", CASES, ids=[f"{n}={v}" for n, v in CASES])
def test_switch_value_matches_the_default(cuda, monkeypatch, fixed_kernel_variants, name, value):
    s = SWITCHES[name]
    scope = s.scope
    if scope == "infer":
        ref = _default("infer", lambda: _infer(cuda))
        monkeypatch.setenv(name, value)
        got = _infer(cuda)
        if s.tol == 0.0:
            _compare(got, ref, 0.0, f"{name}={value}")
        else:
            for k, (a, b) in enumerate(zip(got, ref)):
                e = rel_err(a, b)[1]
                assert e <= s.tol * (10 if k == 2 else 1), (name, value, k, e)          # (k = 2: decoded Detect rows, heavy-tailed)
        return
    if scope == "train_autograd":          # switches of the autograd loss stage: compare inside that stage
        monkeypatch.setenv("YP_NATIVE_STAGE", "0")
        ref = _default("train_autograd", lambda: _train(cuda))
        monkeypatch.setenv(name, value)
        got = _train(cuda)
        _compare(got[0], ref[0], s.tol, f"{name}={value}")
        return
    if scope == "train_adam":
        ref = _default("train_adam", lambda: _train(cuda, steps=2))
        monkeypatch.setenv(name, value)
        got = _train(cuda, steps=2)
        _compare(got[0], ref[0], s.tol, f"{name}={value}")
        return
    if scope == "fp8":
        ref = _default("fp8", lambda: _train(cuda, version="s", fp8=True))
        monkeypatch.setenv(name, value)
        got = _train(cuda, version="s", fp8=True)
        if s.tol == 0.0:
            _compare(got[0], ref[0], 0.0, f"{name}={value}")
            return
        # A component back in bf16: e4m3 / e5m2 rounding is discontinuous, the gradients of this random-weight network decorrelate between ANY
        # two statements that round differently (relative L2 ~1.0 between two PyTorch statements of the fp8 rule) -- what
        # the two runs share is the loss and the size of every gradient.  (Rule-for-rule parity: tests/test_gpu_fp8.py.)
        assert abs(got[1] - ref[1]) <= 0.3 * abs(ref[1]), (name, value, got[1], ref[1])
        ratios = sorted(float(a.norm() / b.norm().clamp_min(1e-30)) for a, b in zip(got[0], ref[0]))
        assert 0.5 <= ratios[len(ratios) // 2] <= 2.0 and ratios[len(ratios) // 10] > 0.2 and ratios[-len(ratios) // 10] < 5.0, (name, value, ratios[::20])
        return
    ref = _default("train", lambda: _train(cuda))
    monkeypatch.setenv(name, value)
    got = _train(cuda)
    if scope == "train_draws":             # another random stream for the InfoNCE sampling: same distribution, other draws
        assert abs(got[1] - ref[1]) <= 0.2 * abs(ref[1]), (name, value, got[1], ref[1])
        errs = sorted(rel_err(a, b)[1] for a, b in zip(got[0], ref[0]))
        assert errs[len(errs) // 2] < 0.5
        return
    _compare(got[0], ref[0], s.tol, f"{name}={value}")


def test_default_tuner_times_its_candidates(cuda, monkeypatch):
    """With no switch set the plan-time autotuner TIMES the applicable kernel variants and keeps the fastest (the registry hands out "" for an
    unset debug switch: a check `is not None` on YP_TUNE_RANDOM once turned the random stress mode on for every run -- configs[1] 0.64 -> 0.86 ms)."""
    from yolopoint_amd import _hip, plan
    monkeypatch.delenv("YP_TUNE_RANDOM", raising=False)
    before = set(plan._TUNE_CACHE)
    pb = plan.PlanBuilder(3, _hip.YP_F16, cuda)
    assert pb.autotune
    x = pb.new_buf(24, 40, 96)
    x.t.normal_()
    g = torch.Generator().manual_seed(5)
    pb.conv(x.view(), torch.randn(160, 96, 3, 3, generator=g) * 0.03, torch.zeros(160), 3, 1, 1, _hip.YP_ACT_SILU)
    pb.finish().run()
    torch.cuda.synchronize()
    new = [v for k, v in plan._TUNE_CACHE.items() if k not in before]
    assert new and all(t is not None and t > 0 for _, t in new), new
