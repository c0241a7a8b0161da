"""Run-to-run reproducibility of the training backward: independently built graphs, same weights and input -> gradient differences
against a reference build with YP_BN_EPILOGUE=0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd.utils.synthetic import make_model, synth_image
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
def run(passes=2):
    m, _ = make_model("n", 3, dtype="bf16")
    m = m.to(dev).train()
    x = synth_image(2, 3, 64, 64, 4).to(dev)
    res = []
    for _ in range(passes):
        m.zero_grad(set_to_none=True)
        o = m(x)
        (o["semi"].square().mean() + o["desc"].mean() + sum(t.tanh().mean() for t in o["objects"])).backward()
        torch.cuda.synchronize()
        res.append(({n: p.grad.clone() for n, p in m.named_parameters()}, {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
    return res
os.environ["YP_BN_EPILOGUE"] = "0"
ref = run()
os.environ["YP_BN_EPILOGUE"] = "1"
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    got = run()
    for ps in range(2):
        g, st = got[ps]
        worst = sorted(((rel(g[n], ref[ps][0][n]), n) for n in g), reverse=True)[:2]
        wst = sorted(((rel(st[n], ref[ps][1][n]), n) for n in st), reverse=True)[:2]
        print(f"trial {trial} pass {ps}: worst grads {[(f'{e:.1e}', n) for e, n in worst]}  worst running stats {[(f'{e:.1e}', n) for e, n in wst]}", flush=True)
