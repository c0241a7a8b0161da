import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from yolopoint_amd import _hip
from yolopoint_amd._hip import lib
from yolopoint_amd.utils.synthetic import planted_predictions
dev = torch.device("cuda:0"); l = lib(); st = _hip.stream_ptr()
S = 1280
nrows = 3 * ((S // 8) ** 2 + (S // 16) ** 2 + (S // 32) ** 2)
for seed in (20, 21):
    p = torch.from_numpy(planted_predictions(1, nrows, 80, 2000, seed, img=S)).to(dev)
    N, nc = p.shape[1], 80
    nb = l.yp_box_nms_workspace_bytes(1, N, nc, 1, 30000)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    boxes = torch.empty((1, 300, 6), device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    def run(max_wh, max_det=300):
        _hip.check(l.yp_box_nms(p.data_ptr(), 1, N, nc, 0.25, 0.45, 1, 1, max_det, 30000, max_wh, boxes.data_ptr(), cnt.data_ptr(), ws.data_ptr(), nb, st))
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    full = t(lambda: run(7680.0)); kept = int(cnt.item()); ncand = int(ws[:4].view(torch.int32).item())
    sort_only = t(lambda: run(-1.0))
    print(f"seed {seed}: rows {N}, candidates {ncand}, kept {kept}: whole call {full:.1f} us, sort only (max_wh<0) {sort_only:.1f} us")
    for md in (50, 100, 200, 300):
        print("   max_det", md, f"{t(lambda: run(7680.0, md)):.1f} us")
if os.environ.get("YP_HIP_LIB", "").endswith("libPB.so"):
    # timeline of thread 0 (wall clock, 100 MHz): pairs (stamp, value)
    cap2 = (nb - 256) // 8
    run(7680.0); torch.cuda.synchronize()
    d = ws[256:].view(torch.int64)[cap2 - 64:cap2].cpu().tolist()
    t0 = d[0]
    print("timeline (us since kernel start, value):", [(round((d[i] - t0) / 100.0, 1), d[i + 1]) for i in range(0, 24, 2) if d[i] >= t0])
