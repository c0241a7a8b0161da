"""Does the fused Adam step bump Tensor._version (TrainGraph.forward keys its filter re-packing on it)?"""
import torch
dev = torch.device("cuda:0")
for fused in (False, True):
    p = torch.nn.Parameter(torch.randn(1000, device=dev))
    opt = torch.optim.Adam([p], lr=1e-3, fused=fused)
    p.grad = torch.randn_like(p)
    v0, before = p._version, p.detach().clone()
    opt.step()
    print(f"fused={fused}: _version {v0} -> {p._version}, changed={not torch.equal(before, p.detach())}")
