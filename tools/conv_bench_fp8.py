#!/usr/bin/env python3
"""Per-layer micro-benchmark of the 8-bit (OCP fp8) convolution kernels on YOLOPoint-l shapes (BASELINE configs[4]: 16 samples/GPU, both
passes of a pair in one launch = 32 images): the 4-wave kernels' non-scaled fp8 MFMAs (tile ids 2, 3, 10-12: bf16 issue rate) against the
8-wave kernel's block-scaled K = 64 MFMAs (tile 57: twice the rate).  Raw C ABI on random bytes; TFLOP/s = 2*M*N*K / time, frac against
the 5 PFLOP/s dense fp8 peak.  python tools/conv_bench_fp8.py [--fmt 0|1] [--tiles 0,3,57]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, YpConvDesc, lib

SHAPES = {   # name: (Cin, Cout, k, s, Hout)
    "c128_128_k3_80": (128, 128, 3, 1, 80), "c256_256_k3_80": (256, 256, 3, 1, 80), "c256_256_k1_80": (256, 256, 1, 1, 80),
    "c256_512_k3s2_40": (256, 512, 3, 2, 40), "c256_256_k3_40": (256, 256, 3, 1, 40), "c512_512_k1_40": (512, 512, 1, 1, 40),
    "c512_1024_k3s2_20": (512, 1024, 3, 2, 20), "c512_512_k3_20": (512, 512, 3, 1, 20), "c1024_1024_k1_20": (1024, 1024, 1, 1, 20),
    "c2048_1024_k1_20": (2048, 1024, 1, 1, 20),
}


def view(t, C_):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], 0, C_, 0
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0,3,57")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--fmt", type=int, default=0)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tiles = [int(t) for t in a.tiles.split(",")]
    fmt = torch.float8_e4m3fn if a.fmt == 0 else torch.float8_e5m2
    print(f"# YOLOPoint-l shapes, batch {a.batch}, activation {'e4m3' if a.fmt == 0 else 'e5m2'} x filter e4m3; columns = tile ids (us | TFLOP/s); frac = best / 5000 TFLOP/s")
    print(f"{'shape':20s} {'M':>7s} {'N':>5s} {'K':>5s} " + " ".join(f"{('t' + str(t)):>14s}" for t in tiles) + f" {'best':>5s} {'TF/s':>7s} {'frac':>6s}")
    for name, (c1, c2, k, s, Ho) in SHAPES.items():
        if a.only and a.only not in name:
            continue
        B, Hi = a.batch, Ho * s
        flat = torch.zeros(B * Hi * Hi * c1 + c1 + 256, dtype=torch.uint8, device=dev)
        x = flat[:B * Hi * Hi * c1].view(B, Hi, Hi, c1)
        x.copy_((torch.randn(B, Hi, Hi, c1, device=dev) * 40).clamp(-448, 448).to(fmt).view(torch.uint8))
        K = k * k * c1
        Kpad = lib().yp_conv_kpad(K, _hip.YP_FP8)
        wq = torch.zeros(c2 + 1, Kpad, dtype=torch.uint8, device=dev)
        wq[:c2, :K] = (torch.randn(c2, K, device=dev) * 40).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        out = torch.zeros(B, Ho, Ho, c2, dtype=torch.bfloat16, device=dev)
        scales = torch.tensor([0.01, 0.001], device=dev)
        res = []
        for tile in tiles:
            d = YpConvDesc()
            d.in0, d.out = view(x, c1), view(out, c2)
            d.weight, d.bias = wq.data_ptr(), None
            d.dtype, d.out_f32, d.B = (_hip.YP_FP8 if a.fmt == 0 else _hip.YP_FP8_BF8), 0, B
            d.Hi, d.Wi, d.Ho, d.Wo = Hi, Hi, Ho, Ho
            d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w = k, k, s, s, k // 2, k // 2
            d.Kpad, d.Npad, d.act, d.tile, d.tail_zero = Kpad, c2, _hip.YP_ACT_NONE, tile, 1
            d.dil_h = d.dil_w = 1
            d.ksplit = 1
            d.scale_in, d.scale_w = scales.data_ptr(), scales.data_ptr() + 4
            st = _hip.stream_ptr()
            if lib().yp_conv2d(C.byref(d), st) != 0:
                res.append(None)
                continue
            for _ in range(3):
                lib().yp_conv2d(C.byref(d), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                lib().yp_conv2d(C.byref(d), st)
            e1.record()
            e1.synchronize()
            res.append(e0.elapsed_time(e1) / a.iters * 1e3)
        M = B * Ho * Ho
        fl = 2.0 * M * c2 * K
        ok = [(u, t) for u, t in zip(res, tiles) if u is not None]
        bu, bt = min(ok)
        tf = lambda u: fl / (u * 1e-6) / 1e12
        cols = " ".join((f"{u:7.1f}|{tf(u):6.0f}" if u is not None else f"{'-':>14s}") for u in res)
        print(f"{name:20s} {M:7d} {c2:5d} {K:5d} {cols} {bt:5d} {tf(bu):7.0f} {tf(bu) / 5000:6.3f}", flush=True)


if __name__ == "__main__":
    main()
