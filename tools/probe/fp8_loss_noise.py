import copy, os, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch
from yolopoint_amd.models.common import invalidate_packed_weights
cuda = torch.device("cuda:0")
m0, _ = make_model("l", 23, dtype="bf16"); m0 = m0.to(cuda).train()
batches = [synthetic_batch(2, 128, cuda, 300 + i) for i in range(3)]
for rep in range(4):
    invalidate_packed_weights()
    m = copy.deepcopy(m0)
    step = TrainStep(m, cuda, img_size=128, lr=1e-3, fp8=True)
    step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
    rows = []
    for it in range(3):
        torch.manual_seed(77 + it)
        step(batches[it])
        rows.append([float(v) for v in step.last_loss_terms.tolist()])
    print(rep, [["%.9g" % v for v in r] for r in rows], flush=True)
