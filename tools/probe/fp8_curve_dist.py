"""Distribution of the statistic of tests/test_gpu_fp8.py::test_fp8_loss_curve_tracks_bf16 over kernel-variant mixtures (YP_TUNE_RANDOM seeds)
and fp8 scale margins: per-step losses of every run -> gpurun_out/fp8_curve_dist.json.  Control: bf16 under another mixture vs bf16 tuned."""
import copy, json, os, sys, time
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import make_model
from yolopoint_amd import plan as yplan
from yolopoint_amd.engine import TrainStep, synthetic_batch
from yolopoint_amd.models.common import invalidate_packed_weights

cuda = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", "200"))
SEEDS = [s for s in os.environ.get("SEEDS", "tuned,1,2,3,4,5,6,7,8,9,10").split(",")]
MARGINS = [float(v) for v in os.environ.get("MARGINS", "1.0,2.0").split(",")]
m0, _ = make_model("l", 11, dtype="bf16")
m0 = m0.to(cuda).train()
batches = [synthetic_batch(2, 128, cuda, 100 + i) for i in range(4)]


def run(fp8, seed, margin):
    yplan._TUNE_CACHE.clear()
    invalidate_packed_weights()
    if seed == "tuned":
        os.environ.pop("YP_TUNE_RANDOM", None)
    else:
        os.environ["YP_TUNE_RANDOM"] = seed
    os.environ["YP_FP8_MARGIN"] = str(margin)
    model = copy.deepcopy(m0)
    step = TrainStep(model, cuda, img_size=128, lr=1e-3, fp8=fp8)
    step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
    losses = []
    t0 = time.time()
    for it in range(STEPS):
        torch.manual_seed(1000 + it)
        losses.append(float(step(batches[it % 4])))
    print(f"fp8={fp8} seed={seed} margin={margin}: {time.time() - t0:.1f}s first20 {sum(losses[:20]) / 20:.4f} last20 {sum(losses[-20:]) / 20:.4f}", flush=True)
    del step, model
    return losses


out = {"bf16": {}, "fp8": {}}
for s in SEEDS:
    out["bf16"][s] = run(False, s, 1.0)
for mg in MARGINS:
    for s in SEEDS:
        out["fp8"][f"{s}@{mg}"] = run(True, s, mg)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/fp8_curve_dist.json", "w"))
ref = out["bf16"][SEEDS[0]]
dev = lambda c, r, n=20: [abs(x - y) / max(abs(y), 1e-6) for x, y in zip(c[:n], r[:n])]
for kind in ("bf16", "fp8"):
    for k, c in out[kind].items():
        d = dev(c, ref)
        tail = sum(c[-20:]) / 20
        print(f"{kind:5s} {k:12s} early max {max(d):.4f} at step {d.index(max(d)):2d} mean {sum(d) / 20:.4f} median {sorted(d)[10]:.4f}  tail {tail:.4f} (ref {sum(ref[-20:]) / 20:.4f})")
