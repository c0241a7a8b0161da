"""Memory-safety check of the plans' kernels by guard bands (what a sanitizer run would look for, on hardware where none runs).

`make -C yolopoint_amd/csrc asan` builds the AddressSanitizer library, but this GPU pool cannot run it: the boxes are in xnack- mode (instrumented
device code objects need xnack+) and the image has no ASan build of the ROCm runtime (the ASan runtime's HSA interceptors abort inside PyTorch's
libamdhip64: profiles/r06_asan_diag.txt).  The substitute that does run everywhere: every activation / gradient buffer
of a plan (plan.Buf) is re-allocated between two 16 KiB CANARY bands, the plans run -- inference (all fusions, ragged non-square input, two
lanes), a bf16 optimizer step, an fp8 optimizer step -- and afterwards
  * both canary bands of every buffer are untouched (no kernel wrote in front of a buffer or behind its tail), and
  * the ZERO TAIL behind every buffer (plan.Buf: one pixel of channels + 64 elements, where the convolutions' DMA fetches out-of-image taps)
    is still all zeros -- the invariant every FAST-path convolution relies on."""
import pytest
import torch

from helpers import make_model
from oracle import net_oracle

pytestmark = pytest.mark.gpu

GUARD = 8192          # elements on each side (16 KiB of 16-bit, 32 KiB of fp32)


@pytest.fixture
def guarded_bufs(monkeypatch):
    from yolopoint_amd import plan
    made = []
    orig = plan.Buf.__init__

    def init(self, B, H, W, C_, tdtype, device, zero=True, storage=None):
        if storage is not None:
            return orig(self, B, H, W, C_, tdtype, device, zero=zero, storage=storage)
        self.B, self.H, self.W, self.C = B, H, W, C_
        n = B * H * W * C_
        size = n + C_ + 64
        canary = 0x5A if tdtype == torch.uint8 else 1024.0          # (exact in f16 / bf16 / f32)
        big = torch.full((GUARD + size + GUARD,), canary, dtype=tdtype, device=device)
        self.flat = big[GUARD:GUARD + size]
        self.flat.zero_()
        self.t = self.flat[:n].view(B, H, W, C_)
        made.append((big, n, size, canary, (B, H, W, C_), tdtype))
    monkeypatch.setattr(plan.Buf, "__init__", init)
    return made


def _check(made, what):
    torch.cuda.synchronize()
    assert len(made) > 10, (what, len(made))
    for big, n, size, canary, shape, tdtype in made:
        front, back, tail = big[:GUARD], big[GUARD + size:], big[GUARD + n:GUARD + size]
        assert bool((front == canary).all()), f"{what}: a kernel wrote IN FRONT of buffer {shape} {tdtype}"
        assert bool((back == canary).all()), f"{what}: a kernel wrote BEHIND the tail of buffer {shape} {tdtype}"
        assert not bool(tail.view(torch.uint8).any()), f"{what}: the zero tail of buffer {shape} {tdtype} was written"


@pytest.mark.parametrize("model_name", ["YOLOPoint", "YOLOPointv52"])
def test_inference_plan_stays_inside_its_buffers(cuda, guarded_bufs, model_name):
    for version, B, H, W, dtype in (("s", 2, 160, 224, "f16"), ("n", 1, 64, 96, "bf16"), ("s", 3, 128, 128, "f32")):
        m, _ = make_model(version, 61, dtype=dtype, model_name=model_name)
        m = m.to(cuda).eval()
        m.fuse()
        x = net_oracle.synth_image(B, 3, H, W, 61).to(cuda)
        with torch.no_grad():
            for _ in range(2):
                o = m(x)
        assert torch.isfinite(o["semi"]).all()
        _check(guarded_bufs, f"{model_name}-{version} {B}x{H}x{W} {dtype}")


@pytest.mark.parametrize("version,fp8", [("n", False), ("s", False), ("s", True)])
def test_training_step_stays_inside_its_buffers(cuda, guarded_bufs, version, fp8):
    from yolopoint_amd.engine import TrainStep, synthetic_batch
    m, _ = make_model(version, 62, dtype="bf16")
    m = m.to(cuda).train()
    step = TrainStep(m, cuda, img_size=128, fp8=fp8)
    step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
    batch = synthetic_batch(2, 128, cuda, 6200)
    for _ in range(2):
        loss = step(batch)
    assert torch.isfinite(loss)
    _check(guarded_bufs, f"train -{version} fp8={fp8}")
