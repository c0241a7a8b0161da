import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import synth_task as T
dev = torch.device("cuda:0")
m = T.train("s", 256, 600, 16, dev, lr=0.004, log=print)
b = T.shapes_batch(16, 256, dev, 77)
with torch.no_grad():
    m.train()
    rm0 = m.model.Conv1.bn.running_mean.clone()
    o_tr = m(b["image"])
    print("running_mean moved by one more forward:", float((m.model.Conv1.bn.running_mean - rm0).abs().max()))
    m.eval()
    o_ev = m(b["image"])
print("train-mode max obj", max(float(t[..., 4].sigmoid().max()) for t in o_tr["objects"]))
pred = o_ev["objects"][0]
print("eval-mode max obj", float(pred[..., 4].max()), "raw eval max obj", max(float(t[..., 4].sigmoid().max()) for t in o_ev["objects"][1]))
for name in ("Conv1", "Conv2", "Conv3"):
    bn = getattr(m.model, name).bn
    print(name, "running_mean", bn.running_mean[:4].tolist(), "running_var", bn.running_var[:4].tolist(), "nbt", int(bn.num_batches_tracked))
# batch statistics of Conv1's raw output for comparison (oracle conv on CPU)
from oracle import net_oracle
sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
x = b["image"].cpu()
raw = torch.nn.functional.conv2d(x, sd["model.Conv1.conv.weight"], None, 2, 2)
print("Conv1 batch mean", raw.mean((0, 2, 3))[:4].tolist(), "batch var", raw.var((0, 2, 3), unbiased=False)[:4].tolist())
ref = net_oracle.yolopoint_forward(sd, x, "s")
print("oracle eval max obj", float(ref["objects"][0][..., 4].max()))
bad = [(k, int((~torch.isfinite(v)).sum())) for k, v in sd.items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
print("non-finite entries:", bad[:10])
print("max |w| per first convs:", [(k, float(v.abs().max())) for k, v in list(sd.items())[:12] if k.endswith("weight")])
neg = [(k, float(v.min())) for k, v in sd.items() if k.endswith("running_var") and float(v.min()) <= 0]
print("non-positive running_var:", neg[:10])
# train-mode oracle forward (batch statistics) on the same weights
import copy
from helpers import NAMES80
from yolopoint_amd import models
mc = models.Model(names=NAMES80, model_name="YOLOPoint", version="s")
