"""Ad-hoc shape sweep: non-square inputs / odd batch sizes through the fp32 and f16 inference plans and the bf16 training graph,
against CPU references built from the same state_dict (no assertions: prints the errors)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_model, rel_err
from oracle import net_oracle
dev = torch.device("cuda:0")
for version, B, H, W in (("n", 3, 96, 160), ("s", 5, 64, 224), ("n", 1, 32, 32), ("s", 7, 160, 96), ("n", 9, 64, 64), ("n", 3, 128, 192), ("s", 5, 64, 192)):
    for dtype in ("f32", "f16"):
        m, sd = make_model(version, 31, dtype=dtype)
        x = net_oracle.synth_image(B, 3, H, W, 5)
        with torch.no_grad():
            ref = net_oracle.yolopoint_forward(sd, x, version)
            got = m.to(dev)(x.to(dev))
        print(f"infer {version} B={B} {H}x{W} {dtype}: semi {rel_err(got['semi'], ref['semi'])[1]:.2e} desc {rel_err(got['desc'], ref['desc'])[1]:.2e} "
              f"pred {rel_err(got['objects'][0], ref['objects'][0])[1]:.2e}", flush=True)
    # training gradients (bf16 graph vs fp32 CPU autograd through the oracle, train-mode BN); sizes must be multiples of 64
    if H % 64 or W % 64:
        continue
    m, sd = make_model(version, 31, dtype="bf16")
    m = m.to(dev).train()
    x = net_oracle.synth_image(B, 3, H, W, 5)
    o = m(x.to(dev))
    (o["semi"].square().mean() + o["desc"].square().mean() + sum(t.square().mean() for t in o["objects"])).backward()
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchors" not in k) for k, v in sd.items()}
    r = net_oracle.yolopoint_forward(p, x, version, training=True)
    (r["semi"].square().mean() + r["desc"].square().mean() + sum(t.square().mean() for t in r["objects"])).backward()
    errs = []
    for (k, v), q in zip(m.state_dict(keep_vars=True).items(), [p[k] for k in sd]):
        if isinstance(v, torch.nn.Parameter) and v.grad is not None and q.grad is not None:
            errs.append((rel_err(v.grad, q.grad)[1], k))
    errs.sort(reverse=True)
    print(f"train {version} B={B} {H}x{W} bf16: worst grad rel-L2 {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs)//2][0]:.2e}, n={len(errs)}", flush=True)
