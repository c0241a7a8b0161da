"""Batch lanes, second look (round 4): the 8 images of configs[1] as n independent sub-batch plans replayed on n streams that were TESTED
to sit on different hardware queues (yp_stream_pick).  The round-4 experiment of DESIGN 4.13 predates that test -- its streams may have
shared a queue.  `python tools/probe/batch_lanes.py [n] [steps]`; YP_INFER_LANES=0: one-lane sub-plans."""
import ctypes as C
import sys
import torch

sys.path.insert(0, ".")
import bench                                   # noqa: E402
from yolopoint_amd import _hip                 # noqa: E402
from yolopoint_amd.utils.synthetic import synth_image   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B, S = 8, 640
    x = synth_image(B, 3, S, S, 1234).to(dev)
    main_s = torch.cuda.Stream(device=dev)
    streams = [main_s]
    for slot in range(1, n):
        out = C.c_void_p()
        _hip.check(_hip.lib().yp_stream_pick(C.c_void_p(main_s.cuda_stream), 1 if slot == 1 else 0, C.byref(out)))
        streams.append(torch.cuda.ExternalStream(out.value, device=dev) if slot <= 2 else torch.cuda.Stream(device=dev))
    models = [bench.build_model("s", "f16", dev)[0] for _ in range(n)]
    b = B // n
    plans = []
    for i, (m, s) in enumerate(zip(models, streams)):
        with torch.cuda.stream(s):
            plans.append(m.model.build_plan(b, S, S, dev, graph=False))
    torch.cuda.synchronize()
    xs = [x[i * b:(i + 1) * b].contiguous() for i in range(n)]

    def step():
        for i in range(1, n):
            streams[i].wait_stream(main_s)
        for i, (m, s) in enumerate(zip(models, streams)):
            with torch.cuda.stream(s):
                m.model.run_plan(plans[i][0], plans[i][1], xs[i])
        for i in range(1, n):
            main_s.wait_stream(streams[i])

    with torch.cuda.stream(main_s):
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
    print(f"batch lanes n={n}: {e0.elapsed_time(e1) / steps:.4f} ms per batch of {B}", flush=True)


if __name__ == "__main__":
    main()
