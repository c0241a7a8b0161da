"""The 16-bit floor of the network (tests/helpers.half_storage_oracle) next to the HIP f16 path, at configs[1] and configs[3]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_model, rel_err, half_storage_oracle, fused_state_dict
from oracle import net_oracle
dev = torch.device("cuda:0")
torch.set_num_threads(32)
for version, B, S, seed in (("s", 8, 640, 1234), ("l", 1, 1280, 77)):
    m, sd = make_model(version, seed, dtype="f16")
    x = net_oracle.synth_image(B, 3, S, S, seed)
    with torch.no_grad():
        ref = net_oracle.yolopoint_forward(sd, x, version)
        with half_storage_oracle(torch.float16):
            flo = net_oracle.yolopoint_forward(fused_state_dict(sd), x, version)
        m = m.to(dev); m.fuse(); m.model.use_graph = True
        m(x.to(dev)); got = m(x.to(dev))
    def rows(o):
        return {"semi": o["semi"], "desc": o["desc"], "pred": o["objects"][0], **{f"raw{i}": t for i, t in enumerate(o["objects"][1])}}
    R, F_, G = rows(ref), rows(flo), rows(got)
    print(f"== YOLOPoint-{version} B={B} {S}x{S}")
    for k in R:
        fm, fl = rel_err(F_[k], R[k]); gm, gl = rel_err(G[k], R[k])
        print(f"  {k:6s} floor max {fm:.3e} l2 {fl:.3e} | hip max {gm:.3e} l2 {gl:.3e} | ratio max {gm/fm:.2f} l2 {gl/fl:.2f}")
    for name, o in (("floor", flo), ("hip", got)):
        a, b = o["semi"].float().cpu().argmax(1, keepdim=True), ref["semi"].argmax(1, keepdim=True)
        bad = a != b
        margin = (ref["semi"].gather(1, b) - ref["semi"].gather(1, a))[bad]
        print(f"  argmax {name}: {int(bad.sum())} of {b.numel()} cells differ, largest reference margin {float(margin.max()) if bad.any() else 0:.3e}, scale {float(ref['semi'].abs().max()):.2f}")
