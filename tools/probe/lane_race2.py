#!/usr/bin/env python3
"""Graph replays with schedule lanes on CHANGING inputs: every replay must equal the eager result for ITS input (a missing dependency
would show the previous input's values)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import make_model
from oracle import net_oracle
cuda = torch.device("cuda:0")
m, _ = make_model("s", 1234, dtype="f16")
xs = [net_oracle.synth_image(8, 3, 640, 640, 100 + i).to(cuda) for i in range(5)]
m = m.to(cuda); m.fuse()
def grab(o):
    d = {k: v.buf.t.clone() for k, v in m.model._dbg_views.items()}
    d.update({"semi": o["semi"].clone(), "desc": o["desc"].clone(), "pred": o["objects"][0].clone(), "x0": o["objects"][1][0].clone(), "x2": o["objects"][1][2].clone()})
    return d
with torch.no_grad():
    os.environ["YP_INFER_LANES"] = "0"
    m.model.use_graph = False
    ref = [grab(m(x)) for x in xs]
    os.environ["YP_INFER_LANES"] = "1"
    m.model._plans.clear()
    for i, x in enumerate(xs):
        o = grab(m(x))
        bad = {k: int((o[k] != ref[i][k]).sum()) for k in o if not torch.equal(o[k], ref[i][k])}
        print("eager lanes", i, "OK" if not bad else f"MISMATCH {bad}")
    m.model.use_graph = True
    for rep in range(2):
        for i, x in enumerate(xs):
            o = grab(m(x))
            bad = {k: int((o[k] != ref[i][k]).sum()) for k in o if not torch.equal(o[k], ref[i][k])}
            lag = {k: torch.equal(o[k], ref[i - 1][k]) for k in bad}
            print(rep, i, "OK" if not bad else f"MISMATCH {bad} equals-previous-input {lag}")
