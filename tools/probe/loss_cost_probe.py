import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch
dev = torch.device("cuda:0")
m, _ = make_model("s", 1, dtype="bf16"); m = m.to(dev).train()
step = TrainStep(m, dev, img_size=640)
batch = synthetic_batch(8, 640, dev, 1234)
def run(n=10):
    for _ in range(3): step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full step: %.1f ms" % run())
real = step.obj_loss
class Fake:
    def build_targets(self, *a, **k): return None
    def __call__(self, p, t, prepared=None): return (sum((x * 1e-9).sum() for x in p), None)
step.obj_loss = Fake()
print("object loss replaced by a 3-op stand-in: %.1f ms" % run())
step.obj_loss = real
import yolopoint_amd.engine as E
real_nce = E.infonce
E.infonce = lambda a, b, *x, **k: (a * 1e-9).sum() + (b * 1e-9).sum()
print("infonce replaced by a stand-in: %.1f ms" % run())
step.obj_loss = Fake()
print("both replaced: %.1f ms" % run())
