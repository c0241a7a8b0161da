"""Which device tensors allocated while a TrainGraph is built are dead afterwards?  A tensor whose pointer went into a plan op must stay alive
(PlanBuilder.keep / TrainGraph.keep); anything listed here with a plan-building call site is a candidate dangling pointer."""
import os, sys, weakref, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_model
from oracle import net_oracle

cuda = torch.device("cuda:0")
dtype = os.environ.get("DT", "f32")
m, sd = make_model("s", 41, dtype=dtype)
m = m.to(cuda).train()
if os.environ.get("FLAT", "1") == "1":
    from yolopoint_amd.dp import GradAllReducer
    from yolopoint_amd.training import grad_ready_groups, link_siblings
    red = GradAllReducer(None, groups=grad_ready_groups(m.model))
    red.flatten_parameters()
    link_siblings(m.model)
    del red
x, xw = net_oracle.synth_image(2, 3, 128, 128, 41).to(cuda), net_oracle.synth_image(2, 3, 128, 128, 42).to(cuda)
records = []
for name in ("zeros", "empty", "ones", "zeros_like", "empty_like", "full", "tensor", "as_strided", "cat", "frombuffer"):
    orig = getattr(torch, name)

    def wrap(*a, _orig=orig, _name=name, **kw):
        t = _orig(*a, **kw)
        if isinstance(t, torch.Tensor) and t.is_cuda:
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "yolopoint_amd" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            records.append((weakref.ref(t), t.data_ptr(), t.numel() * t.element_size(), _name, site))
        return t
    setattr(torch, name, wrap)
out, out_w, heads, graph = m.model.forward_pair(x, xw)
torch.cuda.synchronize()
dead = collections.Counter()
for ref, ptr, nb, name, site in records:
    if ref() is None:
        dead[(name, site)] += 1
print(f"{len(records)} device allocations during build + first forward; dead afterwards by call site:")
for (name, site), n in sorted(dead.items(), key=lambda kv: kv[0][1]):
    print(f"  {n:4d}  torch.{name:10s} {site}")
