import copy, os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import make_model, rel_err
from oracle import net_oracle
from yolopoint_amd.models.common import invalidate_packed_weights
from yolopoint_amd.training import run_native_backward_pair
cuda = "cuda"
ver = os.environ.get("VER", "l")
m, _ = make_model(ver, 7, dtype="bf16")
m = m.to(cuda).train()
m8 = copy.deepcopy(m); m8.model.fp8_train = True
S = int(os.environ.get("S", "128"))
x, xw = net_oracle.synth_image(2, 3, S, S, 7).to(cuda), net_oracle.synth_image(2, 3, S, S, 8).to(cuda)
o, ow, heads, g = m.model.forward_pair(x, xw)
gen = torch.Generator(device=cuda).manual_seed(2)
gs = [torch.randn(t.shape, device=cuda, generator=gen) * 1e-2 for t in heads]
run_native_backward_pair(g, gs[0], gs[1], gs[2:])
for it in range(4):
    for p in m8.parameters(): p.grad = None
    o8, ow8, heads8, g8 = m8.model.forward_pair(x, xw)
    run_native_backward_pair(g8, gs[0], gs[1], gs[2:])
    invalidate_packed_weights()
    errs = sorted((rel_err(p8.grad, p.grad)[1], k) for (k, p), p8 in zip(m.named_parameters(), m8.parameters()))
    print(it, "heads", {k: round(rel_err(o8[k], o[k])[1], 3) for k in ("semi", "desc")}, [round(rel_err(a, b)[1], 3) for a, b in zip(o8["objects"], o["objects"])],
          "grads median/worst", round(errs[len(errs) // 2][0], 3), round(errs[-1][0], 3), errs[-1][1], "best", round(errs[0][0], 3), errs[0][1], "q8", g8.n_q8)
# per-parameter in registration order (sample)
for (k, p), p8 in list(zip(m.named_parameters(), m8.parameters()))[:: max(1, len(list(m.parameters())) // 40)]:
    print(f"{k:50s} {rel_err(p8.grad, p.grad)[1]:.3f}")
st = g8.pack["fp8"]
print("slots", st.n, "scale range", float(st.scale[:st.n].min()), float(st.scale[:st.n].max()))
