#!/usr/bin/env python3
"""Phase clocks of one wave of workgroup 0 of the wave-private split-K kernel (probe build: make -C yolopoint_amd/csrc probewsk;
YP_HIP_LIB=yolopoint_amd/lib/ab/libPW.so).  Prints shader clocks (s_memtime) and the 100 MHz wall clock between the phase stamps."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=256); ap.add_argument("--cout", type=int, default=256); ap.add_argument("--k", type=int, default=3)
ap.add_argument("--s", type=int, default=1); ap.add_argument("--ho", type=int, default=20); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--tile", type=int, default=71); ap.add_argument("--waves", default="0,3,7")
a = ap.parse_args()
dev = torch.device("cuda:0")
pb = PlanBuilder(a.batch, _hip.YP_F16, dev)
pb.autotune = False
Hi = a.ho * a.s
x = pb.new_buf(Hi, Hi, a.cin); x.t.normal_()
w = torch.randn(a.cout, a.cin, a.k, a.k) * (1.0 / (a.cin * a.k * a.k) ** 0.5)
pb.conv(x.view(), w, torch.zeros(a.cout), a.k, a.s, a.k // 2, _hip.YP_ACT_SILU, tile=a.tile)
plan = pb.finish()
L = _hip.lib()
names = {0: "entry", 1: "prologue issued", 12: "loop done", 13: "barrier 1", 14: "partials written + barrier 2", 15: "stores issued"}
for wv in [int(v) for v in a.waves.split(",")]:
    L.yp_debug_wsk_timeline_wave(wv)
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    L.yp_debug_wsk_timeline(buf)
    v = list(buf)
    print(f"-- tile {a.tile} wave {wv}: c{a.cin}->{a.cout} k{a.k} s{a.s} {a.ho}x{a.ho} x{a.batch}")
    prev = None
    for i in range(16):
        if v[i] == 0 and i not in (0,):
            continue
        if prev is not None:
            print(f"   {names.get(prev, 'iter %d' % (prev - 2)):32s} -> {names.get(i, 'iter %d' % (i - 2)):32s} {v[i] - v[prev]:8d} clk")
        prev = i
    print(f"   total {v[15] - v[0]} clk")
    if v[20]:
        print(f"   inside iteration 2: top -> tile landed + fragments in registers {v[20] - v[4]}, addressing of tile it+2 {v[21] - v[20]}, DMA issue + MFMAs {v[22] - v[21]}, loop back {v[5] - v[22]} clk")
