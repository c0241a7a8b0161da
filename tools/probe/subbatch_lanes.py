"""Probe: configs[1] (YOLOPoint-s, 8 x 640 x 640, f16) as N independent sub-batch chains on N stream groups instead of one chain over the
whole batch.  Images are independent through the whole network; the P4 / P5 / PAN part of the forward is ~33 dependent launches whose fixed
cost (boundary + setup + first fetch + epilogue, ~5.5 us) exceeds their multiply-accumulate time, with 50-400 workgroups on 256 CUs.  Several
chains in flight let one chain's fixed-cost phases overlap another's busy phases.  Prints ms per batch of 8 for N = 1, 2, 4, 8."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from yolopoint_amd.utils.synthetic import make_model, synth_image
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B, S = int(os.environ.get("B", "8")), 640
    x = synth_image(B, 3, S, S, 1234).to(dev)
    steps, warm = 200, 30
    for N in (1, 2, 4, 8, 1):
        if B % N:
            continue
        nets, plans, streams, xs = [], [], [], []
        for i in range(N):
            m, _ = make_model("s", 1234, dtype="f16")
            m = m.to(dev)
            m.fuse()
            m.model.static_outputs = True
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                plan, img, outs = m.model.build_plan(B // N, S, S, dev, graph=True)
            nets.append(m.model); plans.append((plan, img, outs)); streams.append(s)
            xs.append(x[i * (B // N):(i + 1) * (B // N)].contiguous())
        root = torch.cuda.Stream(device=dev)

        def step():
            if N == 1:
                with torch.cuda.stream(streams[0]):
                    nets[0].run_plan(plans[0][0], plans[0][1], xs[0])
                return
            ev = root.record_event()
            for i in range(N):
                streams[i].wait_event(ev)
                with torch.cuda.stream(streams[i]):
                    nets[i].run_plan(plans[i][0], plans[i][1], xs[i])
            for i in range(N):
                root.wait_event(streams[i].record_event())
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e3
            best = dt if best is None else min(best, dt)
        print(f"N = {N} sub-batches of {B // N}: {best:.4f} ms per batch of {B} ({B / best * 1e3:.0f} img/s)", flush=True)
        del nets, plans


if __name__ == "__main__":
    main()
