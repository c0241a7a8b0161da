import sys, os
sys.path.insert(0, "/root/repo")
import torch
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch
dev = torch.device("cuda:0")
m, _ = make_model("s", 1, dtype="bf16")
m = m.to(dev).train()
step = TrainStep(m, dev, img_size=320, lr=1e-3)
step.sparse = dict(num_samples_per_image=1000, num_masked_non_matches_per_match=100)
batches = [synthetic_batch(4, 320, dev, 10 + i) for i in range(4)]
hist = []
for it in range(300):
    l = step(batches[it % 4])
    if it % 50 == 0 or it == 299:
        torch.cuda.synchronize()
        hist.append((it, float(l), torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
for h in hist:
    print("step %d loss %.4f allocated %d MB reserved %d MB" % h)
