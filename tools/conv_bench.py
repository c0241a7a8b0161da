#!/usr/bin/env python3
"""Per-layer micro-benchmark of the convolution kernels on the distinct layer shapes of the YOLOPoint family.

  python tools/conv_bench.py --set s8            # YOLOPoint-s, batch 8 (configs[1])
  python tools/conv_bench.py --set l32 --dtype bf16 --tiles 0,3,41,42,43,44 --min-cin 64
      # YOLOPoint-l at 32 images (configs[4]: 16 samples/GPU, both passes of a pair in one launch), compute-bound layers only

Every column is one kernel variant (tile id; 0 = what the plan-time autotuner picks among the 4-wave / halo kernels) timed with HIP events
over `--iters` back-to-back launches on random data; TFLOP/s = 2*M*N*K / time, frac = TFLOP/s / 2500 (dense 16-bit MFMA peak).
Run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...` with --only / one tile for counters."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

# YOLOPoint-s: name: (Cin, Cout, k, s, Hout at 640x640)
SHAPES_S = {
    "stem_3_32_k6s2_320": (3, 32, 6, 2, 320), "c32_64_k3s2_160": (32, 64, 3, 2, 160), "c64_32_k1_160": (64, 32, 1, 1, 160),
    "c32_32_k1_160": (32, 32, 1, 1, 160), "c32_32_k3_160": (32, 32, 3, 1, 160), "c64_64_k1_160": (64, 64, 1, 1, 160),
    "c64_128_k3s2_80": (64, 128, 3, 2, 80), "c128_64_k1_80": (128, 64, 1, 1, 80), "c64_64_k1_80": (64, 64, 1, 1, 80),
    "c64_64_k3_80": (64, 64, 3, 1, 80), "c128_128_k1_80": (128, 128, 1, 1, 80), "c128_128_k3_80": (128, 128, 3, 1, 80),
    "c128_255_k1_80": (128, 255, 1, 1, 80), "c128_256_k3s2_40": (128, 256, 3, 2, 40), "c256_128_k1_40": (256, 128, 1, 1, 40),
    "c128_128_k1_40": (128, 128, 1, 1, 40), "c128_128_k3_40": (128, 128, 3, 1, 40), "c256_256_k1_40": (256, 256, 1, 1, 40),
    "c256_512_k3s2_20": (256, 512, 3, 2, 20), "c512_256_k1_20": (512, 256, 1, 1, 20), "c256_256_k1_20": (256, 256, 1, 1, 20),
    "c256_256_k3_20": (256, 256, 3, 1, 20), "c512_512_k1_20": (512, 512, 1, 1, 20), "c1024_512_k1_20": (1024, 512, 1, 1, 20),
}


def family(scale):
    out = {}
    for name, (c1, c2, k, s, ho) in SHAPES_S.items():
        if c1 <= 4:
            continue
        a, b = c1 * scale, (c2 if c2 in (255,) else c2 * scale)
        out[f"c{a}_{b}_k{k}{'s2' if s == 2 else ''}_{ho}"] = (a, b, k, s, ho)
    return out


SETS = {"s8": (SHAPES_S, 8), "s64": (SHAPES_S, 64), "s16": (SHAPES_S, 16), "l8": (family(2), 8), "l16": (family(2), 16), "l32": (family(2), 32)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="s8", choices=sorted(SETS))
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--min-cin", type=int, default=0)
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--stats", action="store_true", help="training-forward form: raw output + BatchNorm column sums from the epilogue (bn_partial)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    code = _hip.dtype_code(a.dtype)
    shapes, batch = SETS[a.set]
    batch = a.batch or batch
    tiles = [int(t) for t in a.tiles.split(",")]
    print(f"# set {a.set}: batch {batch}, 640x640, {a.dtype}; columns = tile ids (us | TFLOP/s); best = fastest column, frac = best / 2500 TFLOP/s")
    print(f"{'shape':22s} {'M':>7s} {'N':>4s} {'K':>5s} {'GFLOP':>7s} " + " ".join(f"{('t' + str(t)):>14s}" for t in tiles) + f" {'best':>5s} {'TF/s':>7s} {'frac':>6s}")
    tot_fl, tot_us = 0.0, 0.0
    for name, (c1, c2, k, s, Ho) in shapes.items():
        if (a.only and a.only not in name) or c1 < a.min_cin:
            continue
        Hi = Ho * s
        res = []
        rec = None
        for tile in tiles:
            pb = PlanBuilder(batch, code, dev)
            thin = c1 <= 4
            xin = pb.new_buf(Hi, Hi, 4 if thin else c1)
            xin.t.normal_()
            w = torch.randn(c2, c1, k, k) * (1.0 / (c1 * k * k) ** 0.5)
            b = torch.randn(c2) * 0.1
            p = 2 if k == 6 else k // 2
            try:
                extra = None
                if a.stats:
                    M_ = batch * Ho * Ho
                    extra = dict(bn_partial=torch.empty(((M_ + 63) // 64 + 8, 2, (c2 + 63) // 64 * 64 + 64), dtype=torch.float32, device=dev))
                pb.conv(xin.view(), w, None if a.stats else b, k, s, p, _hip.YP_ACT_SILU if (a.act and not a.stats) else _hip.YP_ACT_NONE, tile=tile, extra=extra)
            except _hip.YpError:
                res.append(None)
                continue
            plan = pb.finish()
            st = torch.cuda.Stream()
            try:
                with torch.cuda.stream(st):
                    for _ in range(40):         # (warm: the GPU idled while the plan was built and has dropped its clocks)
                        plan.run()
                    ms = plan.time(a.iters)
            except _hip.YpError:            # (variants that refuse a shape at launch time)
                res.append(None)
                continue
            rec = plan.records[0]
            res.append(ms * 1e3)
            del plan, pb
        if rec is None:
            continue
        ok = [(u, t) for u, t in zip(res, tiles) if u is not None]
        bu, bt = min(ok)
        tf = lambda u: rec.flops / (u * 1e-6) / 1e12
        cols = " ".join((f"{u:7.1f}|{tf(u):6.0f}" if u is not None else f"{'-':>14s}") for u in res)
        print(f"{name:22s} {rec.M:7d} {rec.N:4d} {rec.K:5d} {rec.flops / 1e9:7.2f} {cols} {bt:5d} {tf(bu):7.0f} {tf(bu) / 2500:6.3f}", flush=True)
        tot_fl += rec.flops
        tot_us += bu
    if tot_us:
        print(f"# sum over the listed shapes (one launch each): {tot_fl / 1e9:.1f} GFLOP in {tot_us:.1f} us = {tot_fl / tot_us / 1e6:.0f} TFLOP/s = {tot_fl / tot_us / 1e6 / 2500:.3f} of peak")


if __name__ == "__main__":
    main()
