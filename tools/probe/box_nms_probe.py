"""box NMS kernel time: full vs sort-only (negative max_wh = probe switch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd import _hip
from yolopoint_amd.utils.synthetic import planted_predictions
from yolopoint_amd.utils._ws import workspace
dev = torch.device("cuda:0")
l, st = _hip.lib(), _hip.stream_ptr()
for N, tag in ((25200, "640"), (100800, "1280")):
    pred = torch.from_numpy(planted_predictions(1, N, 80, 2000, 20, img=640)).to(dev)
    out = torch.empty(1, 300, 6, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = workspace(dev, l.yp_box_nms_workspace_bytes(1, N, 80, 1, 30000), "probe")
    for mw in (7680.0, -1.0):
        def run():
            _hip.check(l.yp_box_nms(pred.data_ptr(), 1, N, 80, 0.25, 0.45, 1, 1, 300, 30000, mw, out.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(), st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} {'full' if mw > 0 else 'sort only'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  (kept {int(cnt.item())})")
