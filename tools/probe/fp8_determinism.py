"""Is an fp8 training run bit-reproducible within a process (same weights, batches, draws)?  python tools/probe/fp8_determinism.py [version] [pair]"""
import copy, os, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch
from yolopoint_amd.models.common import invalidate_packed_weights
version, pair = (sys.argv[1] if len(sys.argv) > 1 else "l"), (sys.argv[2] if len(sys.argv) > 2 else "1")
os.environ["YP_TRAIN_PAIR"] = pair
cuda = torch.device("cuda:0")
m0, _ = make_model(version, 23, dtype="bf16"); m0 = m0.to(cuda).train()
batches = [synthetic_batch(2, 128, cuda, 300 + i) for i in range(3)]
def run(env):
    for k, v in env.items(): os.environ[k] = v
    invalidate_packed_weights()
    m = copy.deepcopy(m0)
    step = TrainStep(m, cuda, img_size=128, lr=1e-3, fp8=env.get("FP8", "1") == "1")
    step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
    out = []
    for it in range(3):
        torch.manual_seed(77 + it)
        out.append(float(step(batches[it])))
    return out, [p.detach().clone() for p in m.parameters()], [n for n, _ in m.named_parameters()]
runs = [("twin-only", dict(YP_FP8_TWIN_ONLY="1")), ("twin-only again", dict(YP_FP8_TWIN_ONLY="1")), ("all copies", dict(YP_FP8_TWIN_ONLY="0")), ("all copies again", dict(YP_FP8_TWIN_ONLY="0")),
        ("all copies, 16-bit wgrad", dict(YP_FP8_TWIN_ONLY="0", YP_FP8_WGRAD="0")), ("all copies, 16-bit wgrad again", dict(YP_FP8_TWIN_ONLY="0", YP_FP8_WGRAD="0")),
        ("bf16", dict(FP8="0")), ("bf16 again", dict(FP8="0"))]
res = [(n, run(e)) for n, e in runs]
for (n1, r1), (n2, r2) in zip(res[::2], res[1::2]):
    diff = [nm for a, b, nm in zip(r1[1], r2[1], r1[2]) if not torch.equal(a, b)]
    print(f"{n1:30s} vs {n2:30s}: losses {'equal' if r1[0] == r2[0] else (r1[0], r2[0])}; {len(diff)} of {len(r1[1])} parameters differ {diff[:4]}")
r1, r2 = res[0][1], res[2][1]
diff = [nm for a, b, nm in zip(r1[1], r2[1], r1[2]) if not torch.equal(a, b)]
print(f"twin-only vs all copies: losses {'equal' if r1[0] == r2[0] else (r1[0], r2[0])}; {len(diff)} parameters differ {diff[:6]}")
