"""8-bit (OCP fp8) convolution path of BASELINE.json configs[4]: the quantiser, the e4m3 filter packer and the generic implicit-GEMM
kernel's YP_FP8 (e4m3 x e4m3, forward) / YP_FP8_BF8 (filter e4m3 x dy e5m2, dgrad) instantiations, each against PyTorch on the SAME
8-bit operands (torch.float8_e4m3fn / float8_e5m2 are the OCP formats gfx950 implements): the kernel's only freedom is the fp32
accumulation order and the final bf16 rounding."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from yolopoint_amd import _hip
from yolopoint_amd._hip import YpView, YpConvDesc, lib, check

pytestmark = pytest.mark.gpu
E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2


def padded(B, H, W, C_, dtype, dev):
    """[B,H,W,C] tensor inside a flat buffer followed by zeros (YpConvDesc.tail_zero)."""
    flat = torch.zeros(B * H * W * C_ + C_ + 256, dtype=dtype, device=dev)
    return flat, flat[:B * H * W * C_].view(B, H, W, C_)


def view(t, coff, C_, ups=0):
    v = YpView()
    v.ptr, v.H, v.W, v.cstride, v.coff, v.C, v.ups = t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], coff, C_, ups
    return v


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("B,H,W,Cbuf,coff,C_", [(2, 20, 20, 128, 0, 128), (3, 7, 9, 192, 64, 64), (1, 40, 40, 64, 0, 64)])
def test_quantizer_matches_torch_float8(cuda, fmt, B, H, W, Cbuf, coff, C_):
    torch.manual_seed(H + fmt)
    src = (torch.randn(B, H, W, Cbuf, device=cuda) * 3.0).to(torch.bfloat16)
    src[0, 0, 0, coff] = 1e4                                   # saturates e4m3 (448 * scale) but not e5m2
    dst = torch.zeros(B, H, W, Cbuf, dtype=torch.uint8, device=cuda)
    scale, amax = torch.tensor([0.05], device=cuda), torch.zeros(256, device=cuda)      # (a recorded maximum = 256 sub-slots)
    check(lib().yp_quantize_fp8(view(src, coff, C_), view(dst, coff, C_), _hip.YP_BF16, B, fmt, scale.data_ptr(), amax.data_ptr(), _hip.stream_ptr()))
    torch.cuda.synchronize()
    x = src[..., coff:coff + C_].float() / 0.05
    mx = 448.0 if fmt == 0 else 57344.0
    ref = x.clamp(-mx, mx).to(E4 if fmt == 0 else E5)
    got = dst[..., coff:coff + C_].view(E4 if fmt == 0 else E5)
    assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8))
    assert float(amax.max()) == float(src[..., coff:coff + C_].float().abs().max())
    assert int(dst[..., :coff].sum()) == 0 and int(dst[..., coff + C_:].sum()) == 0      # the other channels of the buffer are untouched
    # the recorded maximum becomes the next scale
    fmax = torch.tensor([mx], device=cuda)
    top = float(amax.max())
    check(lib().yp_fp8_update_scales(scale.data_ptr(), amax.data_ptr(), fmax.data_ptr(), 1, 1.0, _hip.stream_ptr()))
    assert abs(float(scale) - top / mx) < 1e-6 * top / mx and float(amax.abs().max()) == 0.0


def pack_e4m3(w, sw, Kpad, Npad, mode=0, cout_pad=None):
    """torch statement of yp_pack_weight's layouts, quantised: [Npad + 1][Kpad] bytes."""
    Cout, Cin, R, S = w.shape
    q = (w / sw).clamp(-448, 448)
    out = torch.zeros(Npad + 1, Kpad, dtype=torch.uint8, device=w.device)
    if mode == 0:
        rows = q.permute(0, 2, 3, 1).reshape(Cout, R * S * Cin)
        out[:Cout, :R * S * Cin] = rows.to(E4).view(torch.uint8)
    else:
        cp = cout_pad
        t = torch.zeros(Cin, R, S, cp, device=w.device)
        t[..., :Cout] = q.flip(2, 3).permute(1, 2, 3, 0)
        out[:Cin, :R * S * cp] = t.reshape(Cin, R * S * cp).to(E4).view(torch.uint8)
    return out


CONVS = {
    "pointwise_256_256": dict(cin=(256,), cout=256, k=1, s=1, H=20, B=2),
    "pointwise_two_sources_upsampled": dict(cin=(128, 64), ups0=True, cout=192, k=1, s=1, H=16, B=2),
    "conv3x3_residual": dict(cin=(64,), cout=64, k=3, s=1, H=20, B=2, res=True),
    "conv3x3_stride2": dict(cin=(128,), cout=256, k=3, s=2, H=10, B=3),
    "ragged_m_and_n": dict(cin=(192,), cout=72, k=1, s=1, H=9, B=5),
}


@pytest.mark.parametrize("act_fmt", [0, 1])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14])        # 1-5: generic kernel tiles; 10-15: the 3x3 halo kernel (3x3 cases only)
@pytest.mark.parametrize("name", list(CONVS))
def test_fp8_convolution_matches_torch_on_the_same_bytes(cuda, name, tile, act_fmt):
    c = CONVS[name]
    cout, k = c["cout"], c["k"]
    if tile >= 10 and (k != 3 or (tile - 10) % 3 == 2 and cout < 128 or (tile - 10) % 3 == 1 and cout < 64):
        pytest.skip("halo tiles: 3x3 convolutions, tile width <= output channels")
    _check_fp8_conv(cuda, name, c, tile, act_fmt)


# The 8-wave kernel's 8-bit instantiation (csrc/conv_mma8.hip, tile 57): block-scaled K = 64 MFMAs (v_mfma_scale_f32_32x32x64_f8f6f4, unit
# block scales) -- the fp8 forms that issue at TWICE the 16-bit rate; every source a multiple of 128 channels (one 128-byte k row).
CONVS_MX = {
    "pointwise_256_256_ragged_m": dict(cin=(256,), cout=256, k=1, s=1, H=21, B=3),
    "pointwise_k128_single_tile": dict(cin=(128,), cout=136, k=1, s=1, H=12, B=2),
    "pointwise_two_sources_upsampled": dict(cin=(128, 128), ups0=True, cout=192, k=1, s=1, H=16, B=2),
    "conv3x3_residual": dict(cin=(128,), cout=128, k=3, s=1, H=20, B=2, res=True),
    "conv3x3_stride2_deep": dict(cin=(256,), cout=512, k=3, s=2, H=10, B=3),
}


@pytest.mark.parametrize("act_fmt", [0, 1])
@pytest.mark.parametrize("name", list(CONVS_MX))
def test_fp8_mx_kernel_matches_torch_on_the_same_bytes(cuda, name, act_fmt):
    _check_fp8_conv(cuda, name, CONVS_MX[name], 57, act_fmt)


def _check_fp8_conv(cuda, name, c, tile, act_fmt):
    cins, cout, k, s, Ho, B = c["cin"], c["cout"], c["k"], c["s"], c["H"], c["B"]
    Hi = Ho * s
    ups0 = c.get("ups0", False)
    torch.manual_seed(len(name) + tile)
    sx, sw = 0.031, 0.0017
    fmt = E4 if act_fmt == 0 else E5
    srcs, keep = [], []
    for i, ci in enumerate(cins):
        h = Hi >> (1 if (i == 0 and ups0) else 0)
        flat, t = padded(B, h, h, ci, torch.uint8, cuda)
        t.copy_((torch.randn(B, h, h, ci, device=cuda) * 2.0 / sx).clamp(-448, 448).to(fmt).view(torch.uint8))
        srcs.append(t); keep.append(flat)
    w = torch.randn(cout, sum(cins), k, k, device=cuda) * (1.5 / (sum(cins) * k * k) ** 0.5)
    Kpad, Npad = lib().yp_conv_kpad(k * k * sum(cins), _hip.YP_FP8), (cout + 7) // 8 * 8
    wq = pack_e4m3(w, sw, Kpad, Npad)
    out = torch.zeros(B, Ho, Ho, Npad, dtype=torch.bfloat16, device=cuda)
    res = (torch.randn(B, Ho, Ho, Npad, device=cuda)).to(torch.bfloat16) if c.get("res") else None
    scales = torch.tensor([sx, sw], device=cuda)
    d = YpConvDesc()
    d.in0 = view(srcs[0], 0, cins[0], 1 if ups0 else 0)
    if len(srcs) == 2:
        d.in1 = view(srcs[1], 0, cins[1])
    d.out = view(out, 0, Npad)
    if res is not None:
        d.res = view(res, 0, Npad)
    d.weight, d.bias = wq.data_ptr(), None
    d.dtype, d.out_f32, d.B = (_hip.YP_FP8 if act_fmt == 0 else _hip.YP_FP8_BF8), 0, B
    d.Hi, d.Wi, d.Ho, d.Wo = Hi, Hi, Ho, Ho
    d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w = k, k, s, s, k // 2, k // 2
    d.Kpad, d.Npad, d.act, d.tile, d.tail_zero = Kpad, Npad, _hip.YP_ACT_NONE, tile, 1
    d.dil_h = d.dil_w = 1
    d.ksplit = 1
    d.scale_in, d.scale_w = scales.data_ptr(), scales.data_ptr() + 4
    check(lib().yp_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    xs = []
    for i, t in enumerate(srcs):
        x = t.view(fmt).float().permute(0, 3, 1, 2) * sx
        xs.append(F.interpolate(x, scale_factor=2, mode="nearest") if (i == 0 and ups0) else x)
    wr = wq[:cout, :k * k * sum(cins)].view(E4).float().reshape(cout, k, k, sum(cins)).permute(0, 3, 1, 2) * sw
    ref = F.conv2d(torch.cat(xs, 1), wr, None, s, k // 2).permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res[..., :cout].float()
    got = out[..., :cout].float()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, (name, tile, err)                 # one bf16 rounding of the result
    assert float(out[..., cout:].float().abs().max()) == 0.0 if Npad > cout else True


def test_fp8_filter_packer_matches_torch(cuda):
    """yp_pack_weight_fp8_batch: forward (mode 0) and dgrad (mode 1: flipped, channel-transposed) packed e4m3 filters of one master."""
    torch.manual_seed(3)
    Cout, Cin, k = 96, 128, 3
    w = torch.randn(Cout, Cin, k, k, device=cuda) * 0.05
    sw = float(w.abs().max()) / 448.0
    scale, amax = torch.tensor([sw, sw], device=cuda), torch.zeros(2, 256, device=cuda)
    l = lib()
    ents, bufs, blk0 = [], [], 0
    for mode in (0, 1):
        cq, nreal = (Cin, Cout) if mode == 0 else (Cout, Cin)
        Kpad, Npad = l.yp_conv_kpad(k * k * cq, _hip.YP_FP8), (nreal + 7) // 8 * 8
        dst = torch.full((Npad + 1, Kpad), 255, dtype=torch.uint8, device=cuda)
        bufs.append((dst, Kpad, Npad))
        ents.append([w.data_ptr(), dst.data_ptr(), scale.data_ptr() + 4 * mode, amax.data_ptr() + 1024 * mode, Cout, Cin, k, k, 0, Cin, mode, Cout, Kpad, Npad, blk0])
        blk0 += -(-((Npad + 1) * Kpad) // 1024)
    table = torch.tensor(ents, dtype=torch.int64, device=cuda)
    check(l.yp_pack_weight_fp8_batch(table.data_ptr(), 2, blk0, _hip.stream_ptr()))
    torch.cuda.synchronize()
    for mode, (dst, Kpad, Npad) in enumerate(bufs):
        ref = pack_e4m3(w, sw, Kpad, Npad, mode, Cout)
        assert torch.equal(dst, ref), mode
    assert float(amax[0].max()) == float(w.abs().max()) == float(amax[1].max())


def test_fp8_forward_matches_the_fake_quantised_oracle(cuda):
    """Train-mode forward of YOLOPoint-l (every Conv but the stem has channel counts that are multiples of 64) with fp8 Conv operands
    against the oracle with the SAME quantisation rule applied in PyTorch: input and filter of every such Conv rounded to e4m3 with
    per-tensor scale amax / 448 (net_oracle.FAKE_QUANT).  The product's scales lag one pass behind (delayed scaling); on a repeated
    input they equal the current maxima after the calibration passes, so the two differ only by the 16-bit storage between layers --
    the bars are those of the bf16 path, not of fp8 rounding (the fp8-vs-bf16 distance itself is printed: ~0.2-0.3 rel-L2 on this
    random-weight network)."""
    from helpers import make_model, rel_err
    from oracle import net_oracle
    from yolopoint_amd.models.common import invalidate_packed_weights
    m, sd = make_model("l", 7, dtype="bf16")
    m = m.to(cuda).train()
    m.model.fp8_train = True
    x = net_oracle.synth_image(2, 3, 128, 128, 7)
    with torch.no_grad():
        for _ in range(3):                           # scales start at 1: two calibration passes, then the measured one
            out = m(x.to(cuda))
            invalidate_packed_weights()
    graph = next(iter(m.model._train_graphs.values()))[0]
    assert graph.fp8 and graph.n_q8 >= 100, graph.n_q8

    def make_fq(bf16_storage):
        def fq(name, t, w):
            if t.shape[1] % 64:
                return t, w
            if bf16_storage:
                t = t.bfloat16().float()
            q = lambda v: ((v / (v.abs().amax() / 448.0)).clamp(-448, 448).to(E4).float() * (v.abs().amax() / 448.0))
            return q(t), q(w)
        return fq
    refs = []
    with torch.no_grad():
        plain = net_oracle.yolopoint_forward(sd, x, "l", training=True, stats={})
        for bf16_storage in (False, True):
            net_oracle.FAKE_QUANT = make_fq(bf16_storage)
            try:
                refs.append(net_oracle.yolopoint_forward(sd, x, "l", training=True, stats={}))
            finally:
                net_oracle.FAKE_QUANT = None
    ref, ref_b = refs
    # e4m3 rounding is discontinuous: two statements of the SAME quantised network that differ by a 16-bit rounding of the activations
    # between layers (ref vs ref_b, both PyTorch CPU) round a few percent of the elements to different e4m3 values in every layer and end
    # ~0.2-0.3 (rel-L2) apart at the heads of this random-weight, batch-statistics network.  That distance is the resolution of any
    # end-to-end comparison here; the product (bf16 storage between layers) must be as close to the oracle as the oracle's own bf16-storage
    # variant is.  (Exactness of the fp8 kernels themselves: the byte-level tests above.)
    for k in ("semi", "desc"):
        floor, got = rel_err(ref_b[k], ref[k])[1], rel_err(out[k], ref[k])[1]
        print(k, "product vs fake-quantised oracle %.3f | oracle bf16-storage variant vs oracle %.3f | fake-quantised vs plain oracle %.3f"
              % (got, floor, rel_err(ref[k], plain[k])[1]))
        assert got < 1.15 * floor + 1e-2, (k, got, floor)
        cos = torch.nn.functional.cosine_similarity(out[k].flatten().cpu().float(), ref[k].flatten(), dim=0)
        assert float(cos) > 0.9, (k, float(cos))


# The early-step statistic, calibrated on a DISTRIBUTION (tools/probe/fp8_curve_dist.py, gpurun_out/fp8_curve_dist.json: 11 kernel-variant
# mixtures -- the autotuner's picks + YP_TUNE_RANDOM seeds 1..10 -- x {bf16, fp8 margin 1, fp8 margin 2} on one lease, plus the driver's
# round-4 box).  d_t = |loss_x(t) - loss_bf16(t)| / loss_bf16(t) over the first 20 optimizer steps, while the loss falls 13 -> 5:
#   bf16 under ANOTHER variant mixture vs bf16 (the control: same arithmetic, other fp32 summation orders; 20 runs on two leases):
#       max_t d_t 0.10 - 0.49, mean 0.026 - 0.064, median 0.011 - 0.054, tail (last 20 of 200 steps) within 4.9 %
#   fp8 vs bf16 (32 runs + the driver's): max_t d_t 0.09 - 0.31, mean 0.026 - 0.078, median 0.013 - 0.072, tail within 3.0 %
# i.e. the fp8 deviation IS the trajectory noise of this 200-step Adam run at lr 1e-3 (a second bf16 run is as far from the first as the fp8
# run is), and doubling the scale margin (YP_FP8_MARGIN=2: one binade of headroom against a scale that lags the activations) changes
# nothing -- the round-4 failure (a single-step bar of 0.25 derived from ONE run, 0.306 measured on the driver's box) was the tail of that
# noise, not a scale-lag event.  The single-step maximum is therefore only printed; asserted are the robust statistics, at >= 2 x the worst
# of the 53 observations, for the fp8 run AND for the in-test bf16 control (which shows what the bars measure): a broken 8-bit path
# (wrong scale, wrong operand format) is off by O(1) in all of them.
FP8_EARLY_MEAN, FP8_EARLY_MEDIAN, FP8_TAIL = 0.16, 0.15, 0.10


@pytest.mark.statistical
def test_fp8_loss_curve_tracks_bf16(cuda, monkeypatch):
    """200 optimizer steps of the reference training step (engine.TrainStep: both forwards, detector + object + InfoNCE losses, backward,
    Adam) on YOLOPoint-l at 2 x 128 x 128, the same initial weights and the same batches: in bf16, in bf16 under another kernel-variant
    mixture (the control) and with fp8 Conv operands.  The loss curves stay together (mean of the last 20 steps within 10 % of each other,
    mean / median per-step deviation over the first 20 steps inside the bars above) and all come down from where they started."""
    import copy
    from helpers import make_model
    from yolopoint_amd import plan as yplan
    from yolopoint_amd.engine import TrainStep, synthetic_batch
    from yolopoint_amd.models.common import invalidate_packed_weights
    m0, _ = make_model("l", 11, dtype="bf16")
    m0 = m0.to(cuda).train()
    batches = [synthetic_batch(2, 128, cuda, 100 + i) for i in range(4)]     # (a small fixed set: the loss can actually be driven down)
    saved, outer = dict(yplan._TUNE_CACHE), os.environ.get("YP_TUNE_RANDOM")
    curves = {}
    try:
        for name, fp8, mix in (("bf16", False, None), ("bf16 (other variant mixture)", False, "control"), ("fp8", True, None)):
            if mix is not None:
                yplan._TUNE_CACHE.clear()
                monkeypatch.setenv("YP_TUNE_RANDOM", (outer or "") + mix)
            invalidate_packed_weights()
            step = TrainStep(copy.deepcopy(m0), cuda, img_size=128, lr=1e-3, fp8=fp8)
            step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
            losses = []
            for it in range(200):
                torch.manual_seed(1000 + it)             # the InfoNCE sampling of all runs draws the same cells / negatives
                losses.append(float(step(batches[it % 4])))
            curves[name] = losses
            assert all(l == l and abs(l) < 1e6 for l in losses), (name, losses[-5:])
            del step
            if mix is not None:
                if outer is None:
                    monkeypatch.delenv("YP_TUNE_RANDOM")
                else:
                    monkeypatch.setenv("YP_TUNE_RANDOM", outer)
                yplan._TUNE_CACHE.clear()
                yplan._TUNE_CACHE.update(saved)
    finally:
        yplan._TUNE_CACHE.clear()
        yplan._TUNE_CACHE.update(saved)
    b = curves["bf16"]
    head = lambda c: sum(c[:20]) / 20
    tail = lambda c: sum(c[-20:]) / 20
    for name, c in curves.items():
        assert tail(c) < 0.5 * head(c), (name, head(c), tail(c))
        if c is b:
            continue
        early = [abs(x - y) / max(abs(y), 1e-6) for x, y in zip(c[:20], b[:20])]
        mean, median = sum(early) / 20, sorted(early)[10]
        print("%-30s first20 %.4f last20 %.4f (bf16 %.4f / %.4f) | per-step deviation over the first 20 steps: mean %.4f median %.4f max %.4f at step %d"
              % (name, head(c), tail(c), head(b), tail(b), mean, median, max(early), early.index(max(early))))
        assert abs(tail(c) - tail(b)) <= FP8_TAIL * abs(tail(b)), (name, tail(b), tail(c))
        assert mean <= FP8_EARLY_MEAN and median <= FP8_EARLY_MEDIAN, (name, mean, median, early)


# ---------------------------------------------------------------------------------------------
# fp8 GRADIENTS: the dgrad path (e5m2 output gradient x e4m3 flipped filter) end to end
# ---------------------------------------------------------------------------------------------
class _FakeQuantTraining:
    """The oracle with the product's 8-bit rule stated in PyTorch autograd (BASELINE.json configs[4] has no reference implementation; the
    rule is csrc/fp8.hip's): every Conv block whose input channels are a multiple of 64 multiplies e4m3(x) by e4m3(w) (per-tensor scale
    amax / 448); its input gradient is conv_transpose(e5m2(dy), e4m3(w)) (scale amax / 57344) when the OUTPUT channels are a multiple of 64;
    its weight gradient multiplies the same e4m3(x) and e5m2(dy) where both conditions hold (1x1 / 3x3 filters), the 16-bit x and dy elsewhere.  `storage`: None (fp32 between layers) or torch.bfloat16 (the product's storage: each
    convolution's input, output and both gradients rounded to bf16) -- the distance between the two is the resolution of an end-to-end
    comparison, exactly as the 16-bit floor of tests/test_gpu_bench_shapes.py."""

    def __init__(self, storage):
        self.storage, self.flag = storage, [False]

    def __enter__(self):
        import torch.nn.functional as F0
        from oracle import net_oracle
        st, flag = self.storage, self.flag
        rnd = (lambda t: t.to(st).float()) if st is not None else (lambda t: t)
        q4 = lambda v: ((v / (v.abs().amax().clamp_min(1e-30) / 448.0)).clamp(-448, 448).to(E4).float() * (v.abs().amax().clamp_min(1e-30) / 448.0))
        q5 = lambda v: ((v / (v.abs().amax().clamp_min(1e-30) / 57344.0)).clamp(-57344, 57344).to(E5).float() * (v.abs().amax().clamp_min(1e-30) / 57344.0))

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w, stride, pad, quant):
                xs = rnd(x)
                q_fwd, q_bwd = quant and x.shape[1] % 64 == 0, quant and w.shape[0] % 64 == 0
                wq = q4(w) if (q_fwd or q_bwd) else rnd(w)
                ctx.save_for_backward(xs, wq if q_bwd else rnd(w))
                ctx.cfg = (stride, pad, q_bwd, tuple(w.shape), q_fwd and q_bwd and tuple(w.shape[2:]) in ((1, 1), (3, 3)))
                return rnd(F0.conv2d(q4(xs) if q_fwd else xs, wq if q_fwd else rnd(w), None, stride, pad))

            @staticmethod
            def backward(ctx, dy):
                xs, wb = ctx.saved_tensors
                stride, pad, q_bwd, wshape, q_w = ctx.cfg
                dys = rnd(dy)
                dx = torch.nn.grad.conv2d_input(xs.shape, wb, q5(dys) if q_bwd else dys, stride, pad)
                # weight gradient: the same two 8-bit tensors (e4m3 input x e5m2 output gradient) where the layer has both, 16-bit operands elsewhere
                dw = torch.nn.grad.conv2d_weight(q4(xs) if q_w else xs, wshape, q5(dys) if q_w else dys, stride, pad)
                return rnd(dx), dw, None, None, None

        class Shim:
            def __getattr__(self, name):
                return getattr(F0, name)

            @staticmethod
            def conv2d(x, w, b=None, stride=1, padding=0):
                quant, flag[0] = flag[0], False
                y = Fn.apply(x, w, stride, padding, quant)
                return y if b is None else y + b.view(1, -1, 1, 1)

        def mark(name, x, w):
            flag[0] = True
            return x, w
        self.saved = (net_oracle.F, net_oracle.FAKE_QUANT)
        net_oracle.F, net_oracle.FAKE_QUANT = Shim(), mark
        return self

    def __exit__(self, *exc):
        from oracle import net_oracle
        net_oracle.F, net_oracle.FAKE_QUANT = self.saved
        return False


FP8_GRAD = dict(median=1.15, p90=1.15, worst=1.25, worst_abs=0.05, cosine_stable=0.4, cosine_drop_median=0.05, cosine_drop_p90=0.15)


def _fp8_gradient_check(cuda, version, B, S, seed, min_q8, min_stable):
    """All parameter gradients of YOLOPoint-<version> with fp8 Conv operands (forward e4m3 x e4m3, dgrad e5m2 x e4m3, weight gradients e4m3 x e5m2)
    against PyTorch autograd through the oracle with the same rule (_FakeQuantTraining).  e4m3 / e5m2 rounding is discontinuous, so two
    PyTorch statements of the same network that differ only by bf16 storage between layers already end far apart; that distance is the
    floor, and the product -- which IS the bf16-storage variant -- must sit within 1.15x of it on the median and the 90th percentile of
    the per-tensor relative L2 errors (1.25x + 0.05 on the worst tensor); tensors whose direction survives in the floor keep it in the product.
    (Exactness of the dgrad kernels themselves: test_fp8_convolution_matches_torch_on_the_same_bytes / test_fp8_mx_kernel_... with act_fmt = 1.)
    The product's scales are delayed by one pass: three identical forward / backward passes calibrate them to the current maxima."""
    from helpers import make_model, rel_err
    from oracle import net_oracle
    from yolopoint_amd.models.common import invalidate_packed_weights
    m, sd = make_model(version, seed, dtype="bf16")
    m = m.to(cuda).train()
    m.model.fp8_train = True
    x = net_oracle.synth_image(B, 3, S, S, seed)
    grads = {}
    for storage in (None, torch.bfloat16):
        leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
        with _FakeQuantTraining(storage):
            o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
            if storage is None:
                proj = net_oracle.output_projections(o, seed)
            net_oracle.projected_loss(o, proj).backward()
        grads[storage] = leaf
    bn0 = [b.clone() for b in m.buffers()]
    for i in range(3):                                   # passes 1, 2 calibrate the delayed scales (activations, gradients, filters)
        m.zero_grad(set_to_none=True)
        for b, s0 in zip(m.buffers(), bn0):
            b.copy_(s0)
        out = m(x.to(cuda))
        net_oracle.projected_loss(out, proj, cuda).backward()
        invalidate_packed_weights()
    graph = next(iter(m.model._train_graphs.values()))[0]
    assert graph.fp8 and graph.n_q8 >= min_q8, graph.n_q8     # forward AND dgrad convolutions ran on 8-bit operands
    ref, floor_leaf = grads[None], grads[torch.bfloat16]
    hip, floor, cosines = [], [], []
    cosine = lambda a, b: float(torch.nn.functional.cosine_similarity(a.detach().cpu().flatten().double(), b.detach().cpu().flatten().double(), dim=0))
    for name, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        g = ref[name].grad
        hip.append((rel_err(p.grad, g)[1], name))
        floor.append((rel_err(floor_leaf[name].grad, g)[1], name))
        cosines.append((cosine(p.grad, g), cosine(floor_leaf[name].grad, g), name))
    hip.sort(); floor.sort()
    n = len(hip)
    stats = lambda e: (e[n // 2][0], e[int(n * 0.9)][0], e[-1][0])
    (hm, h9, hw), (fm, f9, fw) = stats(hip), stats(floor)
    worst_cos = min(cosines)
    print(f"fp8 gradient rel-L2 vs the fake-quantised oracle: HIP median {hm:.3f} p90 {h9:.3f} worst {hw:.3f} ({hip[-1][1]}) | "
          f"oracle bf16-storage variant median {fm:.3f} p90 {f9:.3f} worst {fw:.3f} ({floor[-1][1]}); lowest HIP cosine {worst_cos[0]:.3f} "
          f"(floor's for that tensor {worst_cos[1]:.3f}, {worst_cos[2]}), lowest floor cosine {min(c[1] for c in cosines):.3f}")
    t = FP8_GRAD
    assert hm <= t["median"] * fm and h9 <= t["p90"] * f9 and hw <= t["worst"] * fw + t["worst_abs"]
    # Direction: 8-bit rounding through ~70 layers decorrelates most gradients of this random-weight network even between the two PyTorch
    # statements (median relative error ~1.0, cosines around 0): there only the error STATISTICS above are comparable.  Where the floor
    # itself keeps the direction (the layers within a few convolutions of the loss), the product must keep it as well.
    stable = [(c_hip, c_floor, name) for c_hip, c_floor, name in cosines if c_floor >= 0.8]
    drop = sorted(c_floor - c_hip for c_hip, c_floor, _ in stable)
    low = min(stable, default=(1.0, 1.0, ""))
    print(f"{len(stable)} of {n} tensors keep their direction in the floor (cosine >= 0.8); lowest HIP cosine among them "
          f"{low[0]:.3f} (floor {low[1]:.3f}, {low[2]}); floor - HIP cosine over them: median {drop[len(drop) // 2] if drop else 0:.3f} "
          f"p90 {drop[int(len(drop) * 0.9)] if drop else 0:.3f} max {drop[-1] if drop else 0:.3f}")
    assert len(stable) >= min_stable
    # Calibrated on 10 kernel-variant mixtures (tools/probe/stat_tests_seeds.sh: the autotuner's picks + YP_TUNE_RANDOM 1..9): the tensor that
    # sits lowest is the same in every run, one whose floor cosine (0.82) is itself marginal, at 0.61 - 0.77 -- the old per-tensor bar
    # (floor - 0.1) was one run's value and failed for 2 of the 10 mixtures.  Asserted now: every stable tensor keeps a clearly positive
    # direction (>= 0.4: 1.5 x the worst observed distance from 1; decorrelated tensors sit around 0), and over the stable set as a whole the
    # product loses no more direction than a few hundredths (median / 90th percentile of floor - HIP).
    for c_hip, c_floor, name in stable:
        assert c_hip >= t["cosine_stable"], (name, c_hip, c_floor)
    assert drop[len(drop) // 2] <= t["cosine_drop_median"] and drop[int(len(drop) * 0.9)] <= t["cosine_drop_p90"], drop
    return graph


@pytest.mark.statistical
def test_fp8_parameter_gradients_against_the_fake_quantised_oracle(cuda):
    """YOLOPoint-s, 4 x 128 x 128 (see _fp8_gradient_check)."""
    _fp8_gradient_check(cuda, "s", 4, 128, 41, min_q8=60, min_stable=4)


@pytest.mark.statistical
def test_fp8_gradients_of_yolopoint_l_through_the_block_scaled_kernels(cuda, monkeypatch):
    """The same check on YOLOPoint-l (configs[4]'s model: every channel count a multiple of 64, most of 128) at 2 x 128 x 128 with the
    autotuner restricted to tile 57 -- the 8-wave kernel on block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 -- so that every forward and
    dgrad convolution it applies to (C % 128 == 0) runs THROUGH it inside a whole training step (the other layers take the library's default
    tile); the byte-level tests cover that kernel in isolation only.  Asserts that those signatures really were served by tile 57."""
    from yolopoint_amd import plan, _hip
    monkeypatch.setattr(plan, "_TUNE_CANDIDATES", (57,))
    saved = dict(plan._TUNE_CACHE)
    plan._TUNE_CACHE.clear()
    try:
        graph = _fp8_gradient_check(cuda, "l", 2, 128, 43, min_q8=100, min_stable=4)
        q8_keys = [k for k in plan._TUNE_CACHE if k[0] in (_hip.YP_FP8, _hip.YP_FP8_BF8)]
        served = [k for k in q8_keys if plan._TUNE_CACHE[k][0] == 57]
        fwd57 = [k for k in served if k[0] == _hip.YP_FP8]
        bwd57 = [k for k in served if k[0] == _hip.YP_FP8_BF8]
        print(f"8-bit convolution signatures: {len(q8_keys)}, served by the block-scaled tile 57: {len(fwd57)} forward (e4m3 x e4m3), {len(bwd57)} dgrad (e5m2 x e4m3)")
        assert len(fwd57) >= 8 and len(bwd57) >= 8, (len(fwd57), len(bwd57), len(q8_keys))
        assert graph.n_q8 >= 100
    finally:
        plan._TUNE_CACHE.clear()
        plan._TUNE_CACHE.update(saved)



@pytest.mark.parametrize("version,pair", [("l", "1"), ("s", "1"), ("s", "0")])
def test_twin_only_batchnorm_outputs_change_nothing(cuda, monkeypatch, version, pair):
    """fp8 mode drops the 16-bit copy of a BatchNorm output that nothing reads (TrainGraph._drop_unread_16bit_copies: the reader analysis over
    the access lists of all plans of the graph).  Skipping a store nobody reads must change NOTHING: three optimizer steps with the copies
    dropped (the default; the unread buffers are NaN-filled, so a reader the analysis missed would poison the run) against YP_FP8_TWIN_ONLY=0
    -- identical losses and bit-identical weights; pair mode and the two-graph mode (YP_TRAIN_PAIR=0: full + keypoint-only backward plans),
    YOLOPoint-l (every channel count a multiple of 64) and -s (8-bit and 16-bit layers mixed); and the analysis really dropped copies."""
    import copy
    from helpers import make_model
    from yolopoint_amd.engine import TrainStep, synthetic_batch
    from yolopoint_amd.models.common import invalidate_packed_weights
    monkeypatch.setenv("YP_TRAIN_PAIR", pair)
    m0, _ = make_model(version, 23, dtype="bf16")
    m0 = m0.to(cuda).train()
    batches = [synthetic_batch(2, 128, cuda, 300 + i) for i in range(3)]
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("YP_FP8_TWIN_ONLY", mode)
        invalidate_packed_weights()
        m = copy.deepcopy(m0)
        step = TrainStep(m, cuda, img_size=128, lr=1e-3, fp8=True)
        step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
        losses = []
        for it in range(3):
            torch.manual_seed(77 + it)
            losses.append(float(step(batches[it])))
        graphs = [g for gs in m.model._train_graphs.values() for g in (gs if isinstance(gs, (list, tuple)) else [gs]) if hasattr(g, "n_twin_only")]
        res[mode] = (losses, [p.detach().clone() for p in m.parameters()], sum(g.n_twin_only for g in graphs), sum(len(g.twin_only) for g in graphs))
        assert all(l == l and abs(l) < 1e6 for l in losses), (mode, losses)
    (l1, p1, dropped, cand), (l0, p0, dropped0, _) = res["1"], res["0"]
    print(f"YOLOPoint-{version} pair={pair}: {dropped} of {cand} twin-writing BatchNorm passes store the twin only")
    assert dropped > 0 and dropped0 == 0
    # (the reported loss VALUE carries one ulp of run-to-run noise in every mode: the object loss adds its per-workgroup sums with a float
    # atomic, csrc/losses.hip::objloss_*_kernel -- tools/probe/fp8_loss_noise.py; gradients and weights do not)
    assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(l1, l0)), (l1, l0)
    names = [n_ for n_, _ in m.named_parameters()]
    bad = [(names[i], float((a - b).abs().max())) for i, (a, b) in enumerate(zip(p1, p0)) if not torch.equal(a, b)]
    if pair == "0":
        # Two-graph mode (YP_TRAIN_PAIR=0, not the default) runs the loss stage through torch autograd (engine._native_stage_ok needs pair mode):
        # its descriptor gradient carries run-to-run fp32 noise (observed 7e-12 on the head gradient, float atomics of the framework's scatter /
        # sampling backward kernels), which now and then flips a 16-bit rounding of ConvDesc's output gradient -- two runs of the SAME twin-only
        # setting differ as often as the two settings do (tools/probe/twin_only_dbg.py: one to two runs in five, |dw| <= 2e-6, tensors of the
        # descriptor head).  A reader of a dropped (NaN-filled) copy would show up as NaN or as an error of the size of the weights.
        bad = [(n_, e_) for n_, e_ in bad if e_ >= 1e-5]
    assert not bad, (len(bad), bad[:6])
