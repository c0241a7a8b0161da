"""Host time of one eager plan replay (enqueue only) against the device time of the same replay: python tools/probe/host_cost.py [batch] [version] [size]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from yolopoint_amd.utils.synthetic import synth_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ver = sys.argv[2] if len(sys.argv) > 2 else "s"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
dev = torch.device("cuda:0")
m, _ = bench.build_model(ver, "f16", dev)
x = synth_image(B, 3, S, S, 1).to(dev)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    plan, img, outs = m.model.build_plan(B, S, S, dev, graph=False)
    for _ in range(20):
        m.model.run_plan(plan, img, x)
    torch.cuda.synchronize()
    hs = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.model.run_plan(plan, img, x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
    print(f"YOLOPoint-{ver} B={B} {S}: host enqueue / replay (ms), total incl. drain (ms):", [(round(a, 3), round(b, 3)) for a, b in hs], "ops", len(plan.records), flush=True)
