#!/usr/bin/env python3
"""Which outputs differ between replays when the inference plan runs its heads on the side lane (YP_INFER_LANES)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import make_model
from oracle import net_oracle
cuda = torch.device("cuda:0")
m, _ = make_model("s", 1234, dtype="f16")
x = net_oracle.synth_image(8, 3, 640, 640, 1234).to(cuda)
m = m.to(cuda); m.fuse()
graph = os.environ.get("LR_GRAPH", "0") == "1"
outs = []
with torch.no_grad():
    for i in range(8):
        m.model.use_graph = graph and i >= 2
        o = m(x)
        torch.cuda.synchronize()
        outs.append({"semi": o["semi"].clone(), "desc": o["desc"].clone(), "pred": o["objects"][0].clone(), "x0": o["objects"][1][0].clone(), "x1": o["objects"][1][1].clone(), "x2": o["objects"][1][2].clone()})
ref = outs[0]
for i, o in enumerate(outs):
    msg = []
    for k in ref:
        d = (o[k] != ref[k])
        n = int(d.sum())
        if n:
            idx = d.nonzero()[:3].tolist()
            msg.append(f"{k}: {n} differ (first {idx})")
    print(i, "graph" if graph and i >= 2 else "eager", msg or "identical to the first (eager) call", float(o["pred"].abs().sum()), float(o["x0"].abs().sum()))
