"""What does a launch pay for following a DIFFERENT kernel?  Phase stamps (probe build, -DYP_TIMELINE) of the same 1x1 convolution A
(256 -> 256 at 20 x 20 x 8, tile 24) (a) launched back to back, (b) behind another instantiation's launch on other buffers (instruction
cache cold for A), (c) behind the SAME instantiation on other buffers (instruction cache warm, data cold), (d) behind a producer that writes A's input.
YP_HIP_LIB=yolopoint_amd/lib/ab/libT.so python tools/probe/cold_start_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yolopoint_amd import _hip
from yolopoint_amd.plan import PlanBuilder
dev = torch.device("cuda:0")
l = _hip.lib(); l.yp_debug_timeline.argtypes = [C.c_void_p]
TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def stamps(build, label):
    pb = PlanBuilder(8, _hip.YP_F16, dev); pb.autotune = False
    build(pb)
    plan = pb.finish()
    for _ in range(30): plan.run()
    torch.cuda.synchronize()
    ms = plan.time(200)
    buf = (C.c_longlong * 64)()
    plan.run(); torch.cuda.synchronize()
    assert l.yp_debug_timeline(buf) == 0
    t = list(buf)
    ks = [i for i in range(2, 40) if t[i] > t[1]]
    print(f"{label:70s} plan {ms * 1e3:6.1f} us | A: setup {t[1] - t[0]:5d}  first k tile +{t[ks[0]] - t[1]:5d}  k loop {t[40] - t[ks[0]]:5d}  epilogue {t[41] - t[40]:5d}  total {t[41] - t[0]:6d} clk")


w = torch.randn(256, 256, 1, 1) * 0.05
w3 = torch.randn(128, 128, 3, 3) * 0.05
def A(pb, x): return pb.conv(x.view() if hasattr(x, "t") else x, w, torch.zeros(256), 1, 1, 0, _hip.YP_ACT_SILU, tile=TILE)

def a_alone(pb):
    x = pb.new_buf(20, 20, 256); x.t.normal_(); A(pb, x)
def other_then_a(pb):
    y = pb.new_buf(40, 40, 128); y.t.normal_()
    pb.conv(y.view(), w3, torch.zeros(128), 3, 1, 1, _hip.YP_ACT_SILU, tile=11)        # 3x3 halo kernel, other buffers
    x = pb.new_buf(20, 20, 256); x.t.normal_(); A(pb, x)
def same_then_a(pb):
    y = pb.new_buf(20, 20, 256); y.t.normal_(); A(pb, y)
    x = pb.new_buf(20, 20, 256); x.t.normal_(); A(pb, x)
def producer_then_a(pb):
    y = pb.new_buf(20, 20, 256); y.t.normal_()
    x = A(pb, y)                                  # same instantiation writes A's input
    A(pb, x)
def other_producer_then_a(pb):
    y = pb.new_buf(20, 20, 128); y.t.normal_()
    x = pb.conv(y.view(), torch.randn(256, 128, 3, 3) * 0.05, torch.zeros(256), 3, 1, 1, _hip.YP_ACT_SILU, tile=11)     # another kernel writes A's input
    A(pb, x)
def many_then_a(pb):
    # a chain of different kernels in front (instruction footprint of a forward), then A on its own input
    y = pb.new_buf(40, 40, 128); y.t.normal_()
    t1 = pb.conv(y.view(), w3, torch.zeros(128), 3, 1, 1, _hip.YP_ACT_SILU, tile=11)
    t2 = pb.conv(t1, torch.randn(128, 128, 1, 1) * 0.05, torch.zeros(128), 1, 1, 0, _hip.YP_ACT_SILU, tile=2)
    t3 = pb.conv(t2, torch.randn(256, 128, 3, 3) * 0.05, torch.zeros(256), 3, 2, 1, _hip.YP_ACT_SILU, tile=12)
    t4 = pb.conv(t3, torch.randn(256, 256, 3, 3) * 0.02, torch.zeros(256), 3, 1, 1, _hip.YP_ACT_SILU, tile=26)
    A(pb, t4)

for fn, label in ((a_alone, "(a) A back to back"), (other_then_a, "(b) another kernel (other buffers), then A"), (same_then_a, "(c) the same instantiation (other buffers), then A"),
                  (producer_then_a, "(d) the same instantiation writes A's input, then A"), (other_producer_then_a, "(e) another kernel writes A's input, then A"),
                  (many_then_a, "(f) four different kernels in a chain, the last writes A's input, then A")):
    stamps(fn, label)
