#!/usr/bin/env python3
"""Achieved HBM bandwidth of the training BatchNorm passes (apply + SiLU; backward = column reduction + fold + row pass) on the tensor shapes of
YOLOPoint-s (16 x 640^2 images per step) and YOLOPoint-l (32 images): algorithmic bytes / HIP-event time per call.
usage: python tools/bn_bench.py [s|l]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, lib, check


def view(t, C_):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], 0, C_, 0
    return v


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "s"
    B = 16 if which == "s" else 32
    w = 1 if which == "s" else 2
    shapes = [(320, 32 * w), (160, 64 * w), (160, 32 * w), (80, 128 * w), (80, 64 * w), (40, 256 * w), (40, 128 * w), (20, 512 * w), (20, 256 * w)]
    dev = torch.device("cuda:0")
    l, st, code = lib(), _hip.stream_ptr(), _hip.YP_BF16
    print(f"# YOLOPoint-{which}, {B} images, bf16: us per call and GB/s of the algorithmic bytes (apply: read x, write y; backward: reduction reads x, dy; row pass reads x, dy, writes dx)")
    print(f"{'H':>4} {'C':>5} {'MB':>7} | {'apply us':>9} {'GB/s':>7} | {'bwd us':>9} {'GB/s':>7}")
    for H, C in shapes:
        raw = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        out, dx = torch.empty_like(raw), torch.empty_like(raw)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
        rmean, rvar = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        nb = l.yp_bn_workspace_bytes(B, H, H, C) + 8 * C + 4096
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        vr, vo, vdy, vdx, none = view(raw, C), view(out, C), view(dy, C), view(dx, C), YpView()
        check(l.yp_bn_stats(vr, code, B, 1e-3, 0.03, mean.data_ptr(), invstd.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), ws.data_ptr(), nb, st))
        ap = timeit(lambda: check(l.yp_bn_act_apply(vr, vo, none, code, B, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, st)))
        bw = timeit(lambda: check(l.yp_bn_act_bwd(vr, vdy, vdx, code, B, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, dg.data_ptr(),
                                                 db.data_ptr(), 0, ws.data_ptr(), nb, st)))
        mb = raw.numel() * 2 / 1e6
        print(f"{H:4d} {C:5d} {mb:7.1f} | {ap:9.1f} {2 * mb / ap * 1e3:7.0f} | {bw:9.1f} {5 * mb / bw * 1e3:7.0f}")


if __name__ == "__main__":
    main()
