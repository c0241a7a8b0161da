"""Times the two InfoNCE gather kernels (csrc/losses.hip: yp_infonce_fwd_grad, yp_infonce_bwd_db) alone on the chip at the sizes of the
training records: -s (8 samples: n = 8 x 3000 matches, 200 negatives, D = 128) and -l (16 samples, D = 256); checks them against the
PyTorch statement of the same sums (fp64) on the way.  `python tools/probe/infonce_bench.py [cap]` (cap = workgroup cap, 0 = none)."""
import sys
import torch

sys.path.insert(0, ".")
from yolopoint_amd import _hip                                  # noqa: E402
from yolopoint_amd.utils.loss_functions import infonce_edges    # noqa: E402


def run(n, negs, D, cap, check):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(n + D)
    dab = torch.nn.functional.normalize(torch.randn((2 * n, D), generator=g), dim=1).to(dev)
    rnd = torch.randint(0, n, (n, negs), generator=g).to(dev)
    idx, order, offsets = infonce_edges(rnd)
    E = idx.shape[1]
    tau = 0.07
    w = torch.empty((n, E), dtype=torch.float32, device=dev)
    rows = torch.empty((n,), dtype=torch.float32, device=dev)
    lse = torch.empty((n,), dtype=torch.float32, device=dev)
    grad = torch.empty_like(dab)
    out = torch.empty_like(dab)
    scale = torch.full((1,), 1.0 / (tau * n), dtype=torch.float32, device=dev)
    pa, pb = dab.data_ptr(), dab.data_ptr() + 4 * n * D
    lib = _hip.lib()

    def fwd():
        _hip.check(lib.yp_infonce_fwd_grad(pa, pb, idx.data_ptr(), n, E, D, 1.0 / tau, w.data_ptr(), rows.data_ptr(), lse.data_ptr(), grad.data_ptr(), None, cap,
                                           _hip.stream_ptr()))

    def bwd():
        _hip.check(lib.yp_infonce_bwd_db(dab.data_ptr(), order.data_ptr(), offsets.data_ptr(), w.data_ptr(), lse.data_ptr(), n, E, D, scale.data_ptr(),
                                         out.data_ptr() + 4 * n * D, None, cap, _hip.stream_ptr()))

    res = {}
    for name, fn in (("fwd_grad", fwd), ("bwd_db", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 100.0          # us per call
    gb = n * E * D * 4 / 1e9
    print(f"n={n} E={E} D={D} cap={cap}: fwd_grad {res['fwd_grad']:.1f} us ({gb / res['fwd_grad'] * 1e6:.0f} GB/s of gathered rows)  "
          f"bwd_db {res['bwd_db']:.1f} us ({gb / res['bwd_db'] * 1e6:.0f} GB/s)", flush=True)
    if check:
        a, b = dab[:n].double(), dab[n:].double()
        m = min(n, 2048)                                  # rows checked for the anchor side
        lg = (a[:m, None, :] * b[idx[:m].long()]).sum(-1) / tau
        p = torch.softmax(lg, 1)
        loss = torch.logsumexp(lg, 1) - lg[:, 0]
        wref = p.clone()
        wref[:, 0] -= 1.0
        dda = (wref[:, :, None] * b[idx[:m].long()]).sum(1)
        e_loss = float((rows[:m].double() - loss).abs().max())
        e_w = float((w[:m].double() - wref).abs().max())
        e_dda = float((grad[:m].double() - dda).abs().max() / dda.abs().max())
        # negative side: ddb[k] = scale * sum_{(i,j): idx[i][j]==k} w[i][j] a[i], all rows (index_add in fp64)
        ddb = torch.zeros((n, D), dtype=torch.float64, device=dev)
        ddb.index_add_(0, idx.long().flatten(), (w.double()[:, :, None] * a[:, None, :]).reshape(-1, D) if n * E * D < 3e8 else
                       torch.zeros((n * E, D), dtype=torch.float64, device=dev))
        if n * E * D < 3e8:
            e_ddb = float((out[n:].double() - ddb * float(scale)).abs().max() / (ddb * float(scale)).abs().max())
        else:
            e_ddb = float("nan")
        print(f"   max err: loss {e_loss:.2e}  w {e_w:.2e}  dda(rel) {e_dda:.2e}  ddb(rel) {e_ddb:.2e}", flush=True)


if __name__ == "__main__":
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    run(2000, 200, 128, cap, True)
    run(2000, 200, 256, cap, True)
    run(24000, 200, 128, cap, False)
    run(48000, 200, 256, cap, False)
    run(24000, 200, 128, 768, False)
    run(48000, 200, 256, 768, False)
