"""Where does a short convolution launch spend its time?  Probe build (make -C yolopoint_amd/csrc probe):
every workgroup records entry / epilogue-done on the 100 MHz wall clock + its HW_ID / XCC_ID.
YP_HIP_LIB=yolopoint_amd/lib/ab/libT.so python tools/probe/wg_census.py
Prints per layer shape: back-to-back launch time, workgroup start spread, lifetime percentiles, end of the last workgroup,
workgroups per CU; plus the time of a plain device copy of the same bytes (the floor a memory-bound layer could reach)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder

dev = torch.device("cuda:0")
l = _hip.lib()
l.yp_debug_wg_times.argtypes = [C.c_void_p, C.c_int]
l.yp_debug_timeline.argtypes = [C.c_void_p]
l.yp_debug_timeline_block.argtypes = [C.c_int]

# (name, Cin, Cout, k, s, Hout, tiles to try)
CASES = [("B1.cv1+cv2 64->64 1x1 160", 64, 64, 1, 1, 160, (1, 2, 4)),
         ("P3 128->128 1x1 80", 128, 128, 1, 1, 80, (2, 3, 4, 6, 7, 8)),
         ("P4 256->256 1x1 40", 256, 256, 1, 1, 40, (2, 3, 4, 6, 7, 8)),
         ("P5 512->512 1x1 20", 512, 512, 1, 1, 20, (3, 4, 6, 7, 8)),
         ("P5 1024->512 1x1 20", 1024, 512, 1, 1, 20, (3, 4, 6, 7, 8)),
         ("Conv5 256->512 3x3s2 20", 256, 512, 3, 2, 20, (3, 4, 12)),
         ("Conv4 128->256 3x3s2 40", 128, 256, 3, 2, 40, (3, 4, 12)),
         ("ConvDesc-like 128->128 3x3 80", 128, 128, 3, 1, 80, (3, 12))]


def copy_floor(nbytes_in, nbytes_out):
    a = torch.empty(nbytes_in // 2, dtype=torch.float16, device=dev).normal_()
    b = torch.empty(nbytes_out // 2, dtype=torch.float16, device=dev)
    n = min(a.numel(), b.numel())
    for _ in range(5):
        b[:n].copy_(a[:n])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        b[:n].copy_(a[:n])
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3, n * 2


def census(name, c1, c2, k, s, Ho, tile):
    Hi = Ho * s
    pb = PlanBuilder(8, _hip.YP_F16, dev); pb.autotune = False
    x = pb.new_buf(Hi, Hi, c1); x.t.normal_()
    try:
        pb.conv(x.view(), torch.randn(c2, c1, k, k) * 0.05, torch.zeros(c2), k, s, k // 2, _hip.YP_ACT_SILU, tile=tile)
    except _hip.YpError as e:
        print(f"  tile {tile}: n/a ({e})"); return
    plan = pb.finish()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(10): plan.run()
        ms = plan.time(100)
        torch.cuda.synchronize()
        n = 16384
        buf = (C.c_ulonglong * (3 * n))()
        zero = (C.c_ulonglong * (3 * n))()
        plan.run(); torch.cuda.synchronize()
        assert l.yp_debug_wg_times(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).copy()
    # workgroups of THIS launch: entries written by it (the array is never cleared; a launch overwrites [0, nwg))
    M = 8 * Ho * Ho
    used = t[:, 0] > 0
    t0 = t[used, 0].astype(np.int64); t1 = t[used, 1].astype(np.int64)
    # keep the entries of the most recent launch: start within 1 ms of the max start
    recent = t0 > t0.max() - 100000
    t0, t1, hw = t0[recent], t1[recent], t[used, 2][recent]
    base = t0.min()
    s_us = (t0 - base) / 100.0; e_us = (t1 - base) / 100.0; life = (t1 - t0) / 100.0
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) * 1000 + (((hw & np.uint64(0xffffffff)) >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
    ncu = len(np.unique(cu))
    pct = lambda a, q: float(np.percentile(a, q))
    print(f"  tile {tile:2d}: {ms*1e3:6.1f} us/launch  wgs {len(t0):5d} on {ncu:3d} CU-ids  start p50/p90/max {pct(s_us,50):5.2f}/{pct(s_us,90):5.2f}/{s_us.max():5.2f}  "
          f"life p10/p50/p90/max {pct(life,10):5.2f}/{pct(life,50):5.2f}/{pct(life,90):5.2f}/{life.max():5.2f}  last end {e_us.max():5.2f} us")


l.yp_debug_probe_mode.argtypes = [C.c_int]
MODES = [int(v) for v in os.environ.get("YP_PROBE_MODES", "0").split(",")]
if os.environ.get("YP_PROBE_CASES"):
    CASES = [c for c in CASES if any(k in c[0] for k in os.environ["YP_PROBE_CASES"].split(","))]
for name, c1, c2, k, s, Ho, tiles in CASES:
    M = 8 * Ho * Ho
    Hi = Ho * s
    bin_, bout = 8 * Hi * Hi * c1 * 2, M * c2 * 2
    cf, nb = copy_floor(bin_, bout)
    print(f"{name}: M={M} in {bin_/1e6:.1f} MB out {bout/1e6:.1f} MB w {c1*c2*k*k*2/1e6:.2f} MB; device copy of {nb/1e6:.1f} MB: {cf:.1f} us")
    for mode in MODES:
        l.yp_debug_probe_mode(mode)
        if len(MODES) > 1:
            print(f" probe mode {mode} (1 = no MFMA, 2 = pixel DMA from the zero page, 4 = filter DMA from the zero row)")
        for tile in tiles:
            census(name, c1, c2, k, s, Ho, tile)
    l.yp_debug_probe_mode(0)
