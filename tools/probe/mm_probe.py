import torch, time
dev = torch.device("cuda:0")
n, D = 12000, 256
a = torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1)
b = torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1)
def timeit(fn, k=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
ref = a.double() @ b.double().t()
g32 = a @ b.t()
print("fp32 mm: %.3f ms, max err %.3e" % (timeit(lambda: a @ b.t()), float((g32.double() - ref).abs().max())))
for dt in (torch.bfloat16, torch.float16):
    ah, bh = a.to(dt), b.to(dt)
    al, bl = (a - ah.float()).to(dt), (b - bh.float()).to(dt)
    try:
        f = lambda: torch.mm(ah, bh.t(), out_dtype=torch.float32) + torch.mm(ah, bl.t(), out_dtype=torch.float32) + torch.mm(al, bh.t(), out_dtype=torch.float32)
        g = f()
        print(dt, "3-split mm out fp32: %.3f ms, max err %.3e" % (timeit(f), float((g.double() - ref).abs().max())))
        A = torch.cat((ah, ah, al), 1); Bc = torch.cat((bh, bl, bh), 1)
        f2 = lambda: torch.mm(A, Bc.t(), out_dtype=torch.float32)
        g2 = f2()
        print(dt, "K-concat single mm: %.3f ms, max err %.3e" % (timeit(f2), float((g2.double() - ref).abs().max())))
    except Exception as e:
        print(dt, "out_dtype unsupported:", repr(e)[:200])
