import torch, ctypes as C
from yolopoint_amd import _hip
from yolopoint_amd._hip import lib, check
torch.manual_seed(0)
n = 1 << 16
dev = "cuda"
p0 = torch.randn(n, device=dev) * 0.1
for fused in (False, True):
    p = p0.clone(); q = torch.nn.Parameter(p0.clone())
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    opt = torch.optim.Adam([q], lr=1e-2, fused=fused, foreach=None if fused else True)
    pd = p0.double().clone(); md = torch.zeros(n, device=dev, dtype=torch.double); vd = md.clone()
    gen = torch.Generator(device=dev).manual_seed(1)
    for it in range(5):
        g = torch.randn(n, device=dev, generator=gen) * (0.1 if it % 2 else 1e-3)
        lr = 1e-2 / (1 + it)
        opt.param_groups[0]["lr"] = lr
        q.grad = g.clone(); opt.step()
        check(lib().yp_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, 0.9, 0.999, 1e-8, 0.0, it + 1, _hip.stream_ptr()))
        gd = g.double()
        md = 0.9 * md + 0.1 * gd; vd = 0.999 * vd + 0.001 * gd * gd
        pd = pd - lr / (1 - 0.9 ** (it + 1)) * md / (vd.sqrt() / (1 - 0.999 ** (it + 1)) ** 0.5 + 1e-8)
        torch.cuda.synchronize()
        print(fused, it, "mine-vs-double", float((p.double() - pd).abs().max()), "torch-vs-double", float((q.data.double() - pd).abs().max()),
              "mine-vs-torch", float((p - q.data).abs().max()))
