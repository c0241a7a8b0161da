"""Is the forward deterministic?  Eager replays against each other, then hipGraph replays against the first eager result.
YOLOPoint-s, batch 8, 640x640, f16 (the benchmarked plan).  Env toggles (YP_FUSE_*, YP_TUNE_ONLY, YP_GRAPH_LINEAR) bisect."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_model
from oracle import net_oracle
dev = torch.device("cuda:0")
B, S = int(os.environ.get("RP_B", "8")), int(os.environ.get("RP_S", "640"))
m, sd = make_model("s", 1234, dtype="f16")
x = net_oracle.synth_image(B, 3, S, S, 1234).to(dev)
m = m.to(dev); m.fuse()
keys = ("semi", "desc")
with torch.no_grad():
    eager = []
    for i in range(5):
        o = m(x)
        eager.append({k: o[k].clone() for k in keys} | {"pred": o["objects"][0].clone()})
    m.model.use_graph = True
    graph = []
    for i in range(5):
        o = m(x)
        graph.append({k: o[k].clone() for k in keys} | {"pred": o["objects"][0].clone()})
    torch.cuda.synchronize()
d = lambda a, b: " ".join("%s %.2e" % (k, float((a[k] - b[k]).abs().max())) for k in a)
print("eager[i] vs eager[0]:", " | ".join(d(e, eager[0]) for e in eager[1:]))
print("graph[i] vs eager[0]:", " | ".join(d(g, eager[0]) for g in graph))
