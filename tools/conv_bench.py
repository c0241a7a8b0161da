#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel on the distinct layer shapes of YOLOPoint-s
 at batch 8, 640x640.  Usage: python tools/conv_bench.py [--only NAME] [--iters N] [--tile T]
Run under `rocprofv3 --pmc ...` for counters."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

# name: (Cin, Cout, k, s, Hout)
SHAPES = {
    "stem_3_32_k6s2_320": (3, 32, 6, 2, 320), "c32_64_k3s2_160": (32, 64, 3, 2, 160), "c64_32_k1_160": (64, 32, 1, 1, 160),
    "c32_32_k1_160": (32, 32, 1, 1, 160), "c32_32_k3_160": (32, 32, 3, 1, 160), "c64_64_k1_160": (64, 64, 1, 1, 160),
    "c64_128_k3s2_80": (64, 128, 3, 2, 80), "c128_64_k1_80": (128, 64, 1, 1, 80), "c64_64_k1_80": (64, 64, 1, 1, 80),
    "c64_64_k3_80": (64, 64, 3, 1, 80), "c128_128_k1_80": (128, 128, 1, 1, 80), "c128_128_k3_80": (128, 128, 3, 1, 80),
    "c128_255_k1_80": (128, 255, 1, 1, 80), "c128_256_k3s2_40": (128, 256, 3, 2, 40), "c256_128_k1_40": (256, 128, 1, 1, 40),
    "c128_128_k1_40": (128, 128, 1, 1, 40), "c128_128_k3_40": (128, 128, 3, 1, 40), "c256_256_k1_40": (256, 256, 1, 1, 40),
    "c256_512_k3s2_20": (256, 512, 3, 2, 20), "c512_256_k1_20": (512, 256, 1, 1, 20), "c256_256_k1_20": (256, 256, 1, 1, 20),
    "c256_256_k3_20": (256, 256, 3, 1, 20), "c512_512_k1_20": (512, 512, 1, 1, 20), "c1024_512_k1_20": (1024, 512, 1, 1, 20),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    code = _hip.dtype_code(a.dtype)
    print(f"{'shape':24s} {'M':>7s} {'N':>4s} {'K':>5s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
    for name, (c1, c2, k, s, Ho) in SHAPES.items():
        if a.only and a.only not in name:
            continue
        Hi = Ho * s
        pb = PlanBuilder(a.batch, code, dev)
        thin = c1 <= 4
        xin = pb.new_buf(Hi, Hi, 4 if thin else c1)
        xin.t.normal_()
        w = torch.randn(c2, c1, k, k) * 0.05
        b = torch.randn(c2) * 0.1
        p = 2 if k == 6 else k // 2
        pb.conv(xin.view(), w, b, k, s, p, _hip.YP_ACT_SILU, tile=a.tile)
        plan = pb.finish()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(5):
                plan.run()
            ms = plan.time(a.iters)
        r = plan.records[0]
        print(f"{name:24s} {r.M:7d} {r.N:4d} {r.K:5d} {ms*1e3:8.1f} {r.flops/(ms*1e-3)/1e12:8.1f} {r.bytes/(ms*1e-3)/1e9:7.0f}")

if __name__ == "__main__":
    main()
