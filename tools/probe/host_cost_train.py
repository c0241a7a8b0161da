"""Host time of one training step (enqueue only, the GPU idle-behind) against the step's device time: python tools/probe/host_cost_train.py [version] [batch]"""
import sys, time
import torch
sys.path.insert(0, ".")
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch

ver = sys.argv[1] if len(sys.argv) > 1 else "s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
m, _ = make_model(ver, 1234, dtype="bf16")
m = m.to(dev).train()
step = TrainStep(m, dev, img_size=640, gas=1)
batch = synthetic_batch(B, 640, dev, 1234)
for _ in range(6):
    step(batch)
torch.cuda.synchronize()
rows = []
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(batch)
    step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((round((t1 - t0) / 2 * 1e3, 3), round((t2 - t0) / 2 * 1e3, 3)))
print(f"YOLOPoint-{ver} B={B}: host ms per step (2 steps enqueued on an idle GPU), ms per step incl. drain:", rows, flush=True)
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
step(batch); step(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
