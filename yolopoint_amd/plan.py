"""Host side of the execution plan: NHWC buffers, channel-slice views, weight packing and the
builder that turns a module tree walk into a native launch list (csrc/plan.hip).

Data layout in HBM: every activation is an NHWC buffer [B, H, W, cstride] in
the compute dtype (f16 / bf16 / f32); a `View` names a channel slice of it, optionally read
through a 2x nearest upsample.  torch.cat never happens: producers write into slices of a
shared buffer, or a conv takes two source views.  Weights are packed once per plan as
[Cout_pad][Kpad] rows with k = (r*S + s)*Cin + c.
"""
import contextlib
import ctypes as C
import os
import math
from .switches import sw

import torch

from . import _hip
from ._hip import YpView, YpConvDesc, YpDetectDesc, YpOpArgs, check, lib


def round_up(v, m):
    return (v + m - 1) // m * m


class Buf:
    """An NHWC device buffer."""

    def __init__(self, B, H, W, C_, tdtype, device, zero=True, storage=None):
        self.B, self.H, self.W, self.C = B, H, W, C_
        # a zero tail behind the pixels: the conv kernels' DMA fetches padding / out-of-image taps from there
        n = B * H * W * C_
        if storage is not None:        # a slice of a caller-owned arena (output-only buffers: no zero tail needed)
            assert storage.numel() >= n and storage.dtype == tdtype
            self.flat = storage
        else:
            self.flat = torch.zeros((n + C_ + 64,), dtype=tdtype, device=device)   # tail >= one pixel of channels + 64
        self.t = self.flat[:n].view(B, H, W, C_)

    def view(self, coff=0, C_=None, ups=0):
        return View(self, coff, self.C - coff if C_ is None else C_, ups)


class View:
    """Channel slice [coff, coff+C) of a Buf; ups=1 reads it through nn.Upsample(2,'nearest')."""

    def __init__(self, buf, coff, C_, ups=0, geom=None):
        self.buf, self.coff, self.C, self.ups = buf, coff, C_, ups
        # geom overrides (H, W, cstride) for the paired-pixel stem view
        self.geom = geom

    @property
    def H(self):
        return (self.geom[0] if self.geom else self.buf.H)

    @property
    def W(self):
        return (self.geom[1] if self.geom else self.buf.W)

    @property
    def cstride(self):
        return (self.geom[2] if self.geom else self.buf.C)

    @property
    def LH(self):   # logical size when read as an input
        return self.H << self.ups

    @property
    def LW(self):
        return self.W << self.ups

    def up(self):
        return View(self.buf, self.coff, self.C, 1, self.geom)

    def slice(self, coff, C_):
        assert coff + C_ <= self.C
        return View(self.buf, self.coff + coff, C_, self.ups, self.geom)

    def c(self):
        return YpView(self.buf.t.data_ptr(), self.H, self.W, self.cstride, self.coff, self.C, self.ups)


NULL_VIEW = YpView(None, 0, 0, 0, 0, 0, 0)


class GeomView:
    """Geometry-only stand-in for an output view (fused Detect decode: nothing is written through it)."""

    def __init__(self, H, W, C_):
        self.H, self.W, self.C = H, W, C_

    def c(self):
        return YpView(None, self.H, self.W, self.C, 0, self.C, 0)


def pack_conv_weight(w, bias, code, device):
    """OIHW fp32 -> [Npad][Kpad] in compute dtype (k = (r*S+s)*Cin + c) + fp32 bias[Npad].

    Pure torch (runs on CPU too, which is how the CPU tests check the layout)."""
    Cout, Cin, R, S = w.shape
    K = R * S * Cin
    Kpad = lib().yp_conv_kpad(K, code)
    Npad = round_up(Cout, 8)
    wk = w.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(Cout, K)
    wp = torch.zeros((Npad + 1, Kpad), dtype=torch.float32, device=wk.device)     # + one zero row (see tail_zero)
    wp[:Cout, :K] = wk
    bp = torch.zeros((Npad,), dtype=torch.float32, device=wk.device)
    if bias is not None:
        bp[:Cout] = bias.detach().to(torch.float32)
    return wp.to(device=device, dtype=_hip.torch_dtype(code)).contiguous(), bp.to(device).contiguous(), Kpad, Npad


def pack_stem_weight(w, bias, code, device):
    """OIHW [Cout, <=4, R, S even] -> the paired-pixel filter [Cout][R][S/2][8] the 16-bit stem kernels read."""
    Cout, Cin, R, S = w.shape
    wpad = torch.zeros((Cout, 4, R, S), dtype=torch.float32, device=w.device)
    wpad[:, :Cin] = w
    wpair = wpad.permute(0, 2, 3, 1).reshape(Cout, R, S // 2, 8).permute(0, 3, 1, 2).contiguous()
    return pack_conv_weight(wpair, bias, code, device)


class MasterWeight:
    """An fp32 OIHW master parameter that the plan itself packs (yp_pack_weight) right before the convolution that
    reads it: training plans re-derive their 16-bit filters on the device every step, with no host work.
    mode 0: forward filter of input channels [c0, c0+cj); mode 1: the dgrad filter (flipped, channel-transposed,
    output channels zero-padded to cout_pad) of the same slice."""

    def __init__(self, param, bias=None, mode=0, c0=0, cj=None, cout_pad=None, q8=False):
        self.param, self.bias, self.mode, self.c0, self.q8 = param, bias, mode, c0, q8      # q8: packed as e4m3 (Fp8State scale of the parameter)
        Cout, Cin, R, S = param.shape
        self.cj = Cin - c0 if cj is None else cj
        self.cout_pad = round_up(Cout, 8) if cout_pad is None else cout_pad
        # logical OIHW shape of the filter the convolution sees
        # (mode "image": the <= 4-channel stem filter; the plan picks the packed form -- pixel pairs for 16-bit plans, 4-channel padding
        # for fp32 ones: yp_pack_weight modes 2 / 3)
        # (mode ("phase", py, px): the dgrad filter of a 3x3 / stride-2 / pad-1 convolution for the input pixels of parity (py, px) -- the
        # (1 + py) x (1 + px) taps that reach them: yp_pack_weight modes 4..7, YpConvDesc.out_phase)
        if isinstance(mode, tuple):
            assert mode[0] == "phase" and (R, S) == (3, 3)
            self.shape = (self.cj, self.cout_pad, 1 + mode[1], 1 + mode[2])
        else:
            self.shape = (Cout, self.cj, R, S) if mode in (0, "image") else (self.cj, self.cout_pad, R, S)


class Fp8State:
    """Per-tensor scales of the 8-bit training convolutions (csrc/fp8.hip): device arrays scale / amax / fmax with one slot per
    quantised tensor (activation buffer, output-gradient buffer, parameter); real = stored * scale.  Quantisation passes read scale[i] and
    record max|x| into amax[i]; update() -- once per optimizer step -- turns the recorded maxima into the next step's scales."""
    FMAX = (448.0, 57344.0)        # e4m3, e5m2

    def __init__(self, device, capacity=8192):
        self.device = device
        self.scale = torch.ones(capacity, dtype=torch.float32, device=device)
        self.amax = torch.zeros(capacity * 256, dtype=torch.float32, device=device)     # 256 sub-slots per tensor (YP_FP8_AMAX_SLOTS)
        self.fmax = torch.full((capacity,), 448.0, dtype=torch.float32, device=device)
        self.n, self.by_key = 0, {}

    def slot(self, key, fmt, init=None):
        if key not in self.by_key:
            assert self.n < self.scale.numel(), "Fp8State: out of scale slots"
            i = self.n
            self.n += 1
            self.by_key[key] = i
            self.fmax[i] = self.FMAX[fmt]
            if init is not None:
                self.scale[i] = max(float(init), 1e-12)
        return self.by_key[key]

    def scale_ptr(self, i):
        return self.scale.data_ptr() + 4 * i

    def amax_ptr(self, i):
        return self.amax.data_ptr() + 1024 * i

    def update(self, margin=1.0):
        if self.n:
            check(lib().yp_fp8_update_scales(self.scale.data_ptr(), self.amax.data_ptr(), self.fmax.data_ptr(), self.n, float(margin), _hip.stream_ptr()))


class OpRecord:
    __slots__ = ("name", "kind", "flops", "bytes", "M", "N", "K")

    def __init__(self, name, kind, flops=0, bytes_=0, M=0, N=0, K=0):
        self.name, self.kind, self.flops, self.bytes, self.M, self.N, self.K = name, kind, flops, bytes_, M, N, K


# conv signature -> fastest kernel/tile id, measured once per process (see PlanBuilder._autotune)
_TUNE_CACHE = {}
# candidate ids: 1..5 generic implicit-GEMM tiles (128x32, 128x64, 128x128, 64x64, 64x32), 21..27 the same tiles with the
# second-generation main loop (several k tiles per barrier, register double-buffered fragments) -- all of these accumulate in the same
# k order, so the output bits do not depend on the tuner's pick; the waves-split-k tiles (31, 33: four interleaved partial sums) are
# reachable by explicit id only; 10..12 the 3x3 halo kernel
# with 32/64/128 output channels per workgroup on 8x16 pixel tiles, 13..15 the same on 4x16 tiles (rejected by the library
# when it does not apply); 16 the persistent fused Bottleneck + C3 tail for C = 32 (csrc/conv_igemm.hip::bneck32_persist_kernel: filters resident,
# next tile's inputs prefetched; bit-identical to tile 10); 17 / 18 / 19 the fused Bottleneck with EIGHT wavefronts per workgroup (BN = 32 / 64 / 128)
# 41..44 / 57: the 8-wave 32x32x16 kernels of csrc/conv_mma8.hip (256x256 / 256x128 / 128x256 / 128x128 tiles; 57 = 256x256 with two-step
# phases, 58 = 256x256 free-running), for 16-bit layers whose channel counts are multiples of 64
# (71..76, the wave-private split-K kernels of csrc/conv_wsk.hip, live in the probe build `make -C yolopoint_amd/csrc probewsk` only since
# round 5: back-to-back they win Conv5 / Conv9 / SPPF.cv2 of configs[1] (22.8 -> 19.2, 16.6 -> 12.1 us), inside the plan -- where every launch also
# pays ~4 us of boundary + prologue + first fetch -- Conv5 26.0 -> 24.6, Conv9 19.4 -> 17.3 and the step 0.6558 -> 0.6535 ms, inside the box noise
# (tools/probe/wsk_ab.sh); as tuner candidates under a random variant mixture they failed the fused-stem equivalence tests: root-caused in round 6 as
# summation order on the plan-unique layers, not corruption -- DESIGN.md 4.13, tests/test_gpu_model.py.)
# 61 / 62: the same kernel with ONE wavefront per SIMD (four waves, a wave = all rows x 64 channels; free-running schedule, one fragment read or DMA
# instruction per MFMA) on 256 / 224 x 256 tiles -- 224 rows make 229 workgroups of M = 51 200 instead of 200 (256 CUs)
_TUNE_CANDIDATES = (1, 2, 3, 4, 5, 21, 22, 23, 24, 25, 26, 27, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 41, 42, 43, 44, 57, 58, 61, 62)
# pixels per BatchNorm-statistics row of the variants whose rows are plain pixel blocks (the 3x3 halo kernels write one row per image tile)
_STAT_ROW_PX = {41: 128, 57: 128, 58: 128, 61: 256, 62: 224}
_TUNE_ITERS = int(sw("YP_TUNE_ITERS"))      # timed launches per candidate
_TUNE_COLD = int(sw("YP_TUNE_COLD"))        # MB swept through the L2s in front of every timed launch (0: back-to-back, hot)
_TUNE_FLUSH = None
if sw("YP_TUNE_ONLY"):           # A/B experiments: restrict the autotuner to a subset of the variants
    _TUNE_CANDIDATES = tuple(int(v) for v in sw("YP_TUNE_ONLY").split(","))


def _shared_tuning_group():
    """(torch.distributed module, rank) when the autotuner's choices are shared across the ranks of a data-parallel job, else None."""
    if sw("YP_TUNE_SHARED") == "0":
        return None
    try:
        import torch.distributed as dist
    except Exception:
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    return dist, dist.get_rank()


_CALLBACK_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class PlanBuilder:
    """Collects launches into a native YpPlan; owns every buffer the plan touches."""

    autotune = True     # time the applicable conv kernel variants on the real buffers when a plan is built

    def __init__(self, B, code, device):
        _hip.require_gpu()
        self.B, self.code, self.device = B, code, device
        self.tdtype = _hip.torch_dtype(code)
        self.ce = 4 if code == _hip.YP_F32 else 8
        self.handle = C.c_void_p()
        check(lib().yp_plan_create(C.byref(self.handle)))
        self.keep = []          # tensors the native plan points into
        self.records = []       # per-op algorithmic work (for roofline accounting)
        self.accesses = []      # per-op (reads, writes) as (buffer key, lo, hi) ranges, for the graph schedule
        self.refreshers = []    # callables that re-pack weights from their (changing) sources: training plans
        self.pack_target = None # (PlanBuilder, cache dict): where MasterWeight pack ops are emitted (None: into this plan)
        self.fp8 = None         # Fp8State of the 8-bit convolutions this plan may hold (TrainGraph sets it)
        self.scope = []

    # -- naming -------------------------------------------------------------------------
    def name(self, leaf=""):
        return ".".join(self.scope + ([leaf] if leaf else []))

    # -- buffers ------------------------------------------------------------------------
    def new_buf(self, H, W, C_, f32=False):
        b = Buf(self.B, H, W, C_, torch.float32 if f32 else self.tdtype, self.device)
        self.keep.append(b.flat)
        return b

    def new_tensor(self, shape, dtype=torch.float32):
        t = torch.zeros(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    # -- dependency tracking (feeds the multi-stream graph schedule) ---------------------------
    @staticmethod
    def _rng(v):
        """An access as (allocation, buffer base, lo, hi, first byte, last byte): two accesses of the SAME buffer conflict when their
        channel (or row) ranges [lo, hi) overlap; accesses through different tensors of one allocation (slices of an arena) conflict
        when their byte ranges do."""
        if isinstance(v, View):
            t = v.buf.t
            lo, hi = (0, 1 << 30) if v.geom is not None else (v.coff, v.coff + v.C)      # (paired-pixel stem view: the whole buffer)
        else:
            t, lo, hi = v                                                                # (tensor, lo, hi) for plain tensors
        b0 = t.data_ptr()
        return (t.untyped_storage().data_ptr(), b0, lo, hi, b0, b0 + t.numel() * t.element_size())

    def _track(self, reads, writes):
        self.accesses.append(([self._rng(v) for v in reads if v is not None], [self._rng(v) for v in writes if v is not None]))

    def dependencies(self):
        """deps[j] = sorted earlier op indices j must wait for (RAW, WAR, WAW on overlapping slices)."""
        def overlap(a, b):
            if a[0] != b[0]:
                return False
            if a[1] == b[1]:
                return a[2] < b[3] and b[2] < a[3]
            return a[4] < b[5] and b[4] < a[5]
        deps = []
        for j, (rd, wr) in enumerate(self.accesses):
            d = set()
            for i in range(j):
                ri, wi = self.accesses[i]
                if any(overlap(r, w) for r in rd for w in wi) or any(overlap(w, x) for w in wr for x in ri + wi):
                    d.add(i)
            deps.append(sorted(d))
        # transitive reduction light: a dependency already implied by another dependency of j adds an edge and nothing else
        reach = []
        for j, d in enumerate(deps):
            r = set()
            for i in d:
                r |= reach[i]
            keep = [i for i in d if i not in r]
            deps[j] = keep
            reach.append(r | set(d))
        return deps

    # -- ops ----------------------------------------------------------------------------
    def conv(self, srcs, w, bias, k, s, p, act, out=None, res=None, out_f32=False, tile=0, out2=None, detect=None, extra=None):
        """srcs: one or two Views (channel-concatenated); w: OIHW fp32 tensor (BN already folded), or a callable
        returning (w, bias) — then the packed copy is re-derived by refresh() before every training step.
        k/s/p: square kernel, stride, padding.  extra: raw descriptor overrides used by dgrad / wgrad:
          out_hw=(Ho,Wo), dil=int, zero_stuffed=bool (in0 read as a zero-stuffed 2x tensor), ksplit=int,
          kernel_hw=(R,S) for non-square "filters", raw_weight=(tensor [Npad+1][Kpad], Kpad, Npad) to bypass packing,
          pre=(w1, b1, act1): fused Bottleneck — this 3x3 reads act1(conv1x1(src, w1) + b1), the hidden tensor stays in LDS,
          bn_partial=tensor: the generic kernel's epilogue also writes the per-64-pixel-block column sums (yp_bn_finalize),
          dry_run=True: build + autotune the launch but do not add it; returns the measured ms per launch (None if untimed)."""
        if isinstance(srcs, View):
            srcs = [srcs]
        assert 1 <= len(srcs) <= 2
        extra = extra or {}
        w_fn = None
        master = w if isinstance(w, MasterWeight) else None
        if master is not None:
            bias = master.bias
        elif callable(w):
            w_fn = w
            w, bias = w_fn()
        raw_weight = extra.get("raw_weight")
        if raw_weight is not None:
            Cout, Cin = extra["cout"], sum(v.C for v in srcs)
            R, S = extra["kernel_hw"]
        elif master is not None:
            Cout, Cin, R, S = master.shape
        else:
            Cout, Cin, R, S = w.shape
        sh = sw = s
        ph = pw = p
        dil = extra.get("dil", 1)
        v0 = srcs[0]
        zs = bool(extra.get("zero_stuffed", False))
        if zs:
            v0 = v0.up()
            srcs = [v0] + list(srcs[1:])
        Hi, Wi = v0.LH, v0.LW
        Ho = (Hi + 2 * ph - dil * (R - 1) - 1) // sh + 1
        Wo = (Wi + 2 * pw - dil * (S - 1) - 1) // sw + 1
        if "out_hw" in extra:
            Ho, Wo = extra["out_hw"]
        q8 = extra.get("q8")                        # dict(dtype=YP_FP8 | YP_FP8_BF8, slot=Fp8State slot of the (common) input scale): srcs are 1-byte views
        conv_dtype = q8["dtype"] if q8 else self.code
        thin = q8 is None and raw_weight is None and len(srcs) == 1 and v0.C == 4 and v0.cstride == 4 and Cin <= 4
        pair = thin and self.ce == 8
        if pair:
            # 16-bit stem (6x6/s2/p2, models/YOLOPoint.py:156): pair adjacent pixels -> view [H, W/2, 8]; taps pair up along s
            if S % 2 or sw % 2 or pw % 2 or Wi % 2:
                raise _hip.YpError("thin-input conv needs even kernel width / stride / pad for 16-bit dtypes")
            v0 = View(v0.buf, 0, 8, 0, geom=(v0.H, v0.W // 2, 8))
            srcs = [v0]
            S, sw, pw, Wi = S // 2, sw // 2, pw // 2, Wi // 2
        elif not thin:
            assert sum(v.C for v in srcs) == Cin, (self.name(), [v.C for v in srcs], Cin)

        def prep(w_, b_):
            """master OIHW weights -> packed device copy (image-like inputs: pad to 4 channels, pair pixels)."""
            if thin:
                wpad = torch.zeros((w_.shape[0], 4, w_.shape[2], w_.shape[3]), dtype=torch.float32, device=w_.device)
                wpad[:, :w_.shape[1]] = w_
                w_ = wpad
                if pair:
                    w_ = w_.permute(0, 2, 3, 1).reshape(w_.shape[0], w_.shape[2], w_.shape[3] // 2, 8).permute(0, 3, 1, 2).contiguous()
            return pack_conv_weight(w_, b_, self.code, self.device)

        Cout_pad = round_up(Cout, 8)
        if detect is not None:
            # fused Detect decode: the conv writes x_out / z directly, `out` only carries the geometry
            assert out is None and out2 is None and res is None and out_f32
            out = GeomView(Ho, Wo, Cout_pad)
        elif out is None:
            assert out2 is None
            out = self.new_buf(Ho, Wo, Cout_pad, f32=out_f32).view()
        c2 = out2.C if out2 is not None else 0      # channels [out.C, out.C + c2) are written to out2
        post = extra.get("post")                    # fused C3 tail: `out` is the tail's destination (2 * Cout channels)
        assert out.C + c2 == (2 * Cout_pad if post is not None else Cout_pad) and out.H == Ho and out.W == Wo, (self.name(), out.C, c2, Cout_pad, out.H, Ho)
        if raw_weight is not None:
            wp, Kpad, Npad = raw_weight
            bp = None
            self.keep += [wp]
        elif master is not None:
            assert thin == (master.mode == "image"), "image-like (thin) inputs take MasterWeight(mode='image')"
            pmode = (2 if pair else 3) if thin else (4 + 2 * master.mode[1] + master.mode[2] if isinstance(master.mode, tuple) else master.mode)
            # (thin: 4 k slots per filter pixel -- S was halved above when pixels are paired and each tap covers 8)
            Kpad, Npad = lib().yp_conv_kpad(R * S * ((8 if pair else 4) if thin else Cin), conv_dtype), round_up(Cout, 8)
            assert bool(master.q8) == (q8 is not None)
            # the pack op goes into `pack_target` (a builder shared by every plan over the same parameters, replayed once per
            # optimizer step) when one is set, else in front of the convolution in this plan
            tgt, cache = self.pack_target if self.pack_target is not None else (self, None)
            key = (master.param.data_ptr(), master.bias.data_ptr() if master.bias is not None else 0, pmode, master.c0, master.cj, master.cout_pad,
                   bool(master.q8))
            w_slot = None
            if master.q8:
                # e4m3 packed copy: one scale slot per parameter (its forward and dgrad copies hold the same values)
                w_slot = self.fp8.slot(("w", master.param.data_ptr()), 0, init=float(master.param.detach().abs().max()) / 448.0)
            if cache is not None and key in cache:
                wp, bp = cache[key]
            elif master.q8:
                assert master.bias is None
                wp = torch.zeros((Npad + 1, Kpad), dtype=torch.uint8, device=self.device)
                bp = torch.zeros((Npad,), dtype=torch.float32, device=self.device)
                mo, mi, mr, ms = master.param.shape
                tgt.keep += [wp, bp, master.param]
                tgt.__dict__.setdefault("pack8_entries", []).append(
                    [master.param.data_ptr(), wp.data_ptr(), self.fp8.scale_ptr(w_slot), self.fp8.amax_ptr(w_slot), mo, mi, mr, ms,
                     master.c0, master.cj, pmode, master.cout_pad, Kpad, Npad])
                if cache is not None:
                    cache[key] = (wp, bp)
            else:
                wp = torch.zeros((Npad + 1, Kpad), dtype=self.tdtype, device=self.device)
                bp = torch.zeros((Npad,), dtype=torch.float32, device=self.device)
                mo, mi, mr, ms = master.param.shape
                tgt.keep += [wp, bp, master.param] + ([master.bias] if master.bias is not None else [])
                tgt.op(_hip.OP_PACK_WEIGHT, [], [(wp, 0, 1 << 30)], "pack_w", f=[master.param, master.bias], g=[bp], p=[wp],
                       i=[self.code, mo, mi, mr, ms, master.c0, master.cj, pmode], n=[Kpad, Npad | (master.cout_pad << 32)])
                # (the same arguments as a row of the batched packer's table: yp_pack_weight_batch, TrainGraph.forward)
                tgt.__dict__.setdefault("pack_entries", []).append(
                    [master.param.data_ptr(), wp.data_ptr(), master.bias.data_ptr() if master.bias is not None else 0, bp.data_ptr(), mo, mi, mr, ms,
                     master.c0, master.cj, pmode, master.cout_pad, Kpad, Npad])
                if cache is not None:
                    cache[key] = (wp, bp)
            self.keep += [wp, bp]
            extra = dict(extra, weight_view=(wp, 0, 1 << 30)) if "weight_view" not in extra else extra
        else:
            wp, bp, Kpad, Npad = prep(w, bias)
            self.keep += [wp, bp]
            if w_fn is not None:
                def refresh(wp=wp, bp=bp, w_fn=w_fn, prep=prep):
                    w2, b2 = w_fn()
                    nw, nb, _, _ = prep(w2, b2)
                    wp.copy_(nw)
                    bp.copy_(nb)
                self.refreshers.append(refresh)
        d = YpConvDesc()
        d.in0 = v0.c()
        d.in1 = srcs[1].c() if len(srcs) == 2 else NULL_VIEW
        d.out = out.c()
        d.res = res.c() if res is not None else NULL_VIEW
        d.out2 = out2.c() if out2 is not None else NULL_VIEW
        d.weight = wp.data_ptr()
        d.bias = bp.data_ptr() if (bias is not None and bp is not None) else None
        d.dtype, d.out_f32, d.B = conv_dtype, int(out_f32), extra.get("batch", self.B)
        if q8 is not None:
            assert master is not None and w_slot is not None
            d.scale_in, d.scale_w = self.fp8.scale_ptr(q8["slot"]), self.fp8.scale_ptr(w_slot)
        d.Hi, d.Wi, d.Ho, d.Wo = Hi, Wi, Ho, Wo
        d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w = R, S, sh, sw, ph, pw
        d.dil_h = d.dil_w = dil
        d.in0_zero_stuffed = int(zs)
        d.ksplit = int(extra.get("ksplit", 1))
        slabs = extra.get("split_slabs")            # deterministic split-K: (tensor, stride in floats)
        if slabs is not None:
            d.split_slabs, d.split_stride = slabs[0].data_ptr(), int(slabs[1])
            self.keep.append(slabs[0])
        d.atomic_accumulate = int(extra.get("atomic", 0))
        if extra.get("out_phase") is not None:      # (py, px): `out` / `res` are that parity class of a [2 Ho][2 Wo] tensor (views with geom = (Ho, Wo, cstride))
            d.out_phase = 1 + 2 * int(extra["out_phase"][0]) + int(extra["out_phase"][1])
        d.Kpad, d.Npad, d.act, d.tile, d.tail_zero = Kpad, Npad, act, tile, 1
        bn_partial = extra.get("bn_partial")        # fp32 [ceil(M/64)][2][Cout_pad]: BatchNorm column sums written by the epilogue
        if bn_partial is not None:
            d.bn_partial = bn_partial.data_ptr()
            self.keep.append(bn_partial)
        pre = extra.get("pre")
        if pre is not None:
            w1, b1, act1 = pre
            assert w1.shape[2] == w1.shape[3] == 1 and w1.shape[0] == w1.shape[1] == Cin == Cout, (self.name(), tuple(w1.shape), Cin, Cout)
            wp1, bp1, Kpad1, Npad1 = pack_conv_weight(w1, b1, self.code, self.device)
            self.keep += [wp1, bp1]
            d.pre_weight, d.pre_bias = wp1.data_ptr(), (bp1.data_ptr() if b1 is not None else None)
            d.pre_Kpad, d.pre_Npad, d.pre_act = Kpad1, Npad1, act1
        if post is not None:
            w3, b3, act3, u = post
            assert pre is not None and len(srcs) == 1 and tuple(w3.shape) == (2 * Cout, 2 * Cout, 1, 1) and u.C == Cout, (self.name(), tuple(w3.shape), u.C, Cout)
            wp3, bp3, Kpad3, Npad3 = pack_conv_weight(w3, b3, self.code, self.device)
            self.keep += [wp3, bp3]
            d.in1 = u.c()
            d.post_weight, d.post_bias = wp3.data_ptr(), (bp3.data_ptr() if b3 is not None else None)
            d.post_Kpad, d.post_Npad, d.post_act = Kpad3, Npad3, act3
        det = None
        if detect is not None:
            det = YpDetectDesc()
            det.na, det.no, det.stride = detect["na"], detect["no"], float(detect["stride"])
            for i, v in enumerate(detect["anchors_px"]):
                det.anchors_px[i] = float(v)
            det.x_out = detect["x_out"].data_ptr()
            det.z_out = detect["z_out"].data_ptr() if detect["z_out"] is not None else None
            det.rows_total, det.row_offset = detect["rows_total"], detect["row_offset"]
        tuned_ms = None
        if tile == 0 and self.autotune and d.ksplit == 1 and not d.atomic_accumulate:
            d.tile, tuned_ms = self._autotune(d, det, (conv_dtype, d.B, Hi, Wi, tuple((v.C, v.ups) for v in srcs), Cout_pad, R, S, sh, sw, dil, zs,
                                                       int(out_f32), res is not None, c2, act, detect is not None, pre is not None, post is not None,
                                                       bn_partial is not None, extra.get("stat_group_px"), extra.get("out_phase")), stat_group_px=extra.get("stat_group_px"))
        if bn_partial is not None:                  # how many partial rows the chosen kernel variant writes (they are batch-major)
            rows = C.c_int(0)
            check(lib().yp_conv_bn_partial_rows(C.byref(d), C.byref(rows)))
            self.last_bn_rows = rows.value
            assert rows.value <= bn_partial.shape[0], (self.name(), rows.value, tuple(bn_partial.shape))
        if extra.get("dry_run"):
            return tuned_ms
        if det is not None:
            check(lib().yp_plan_add_conv_detect(self.handle, C.byref(d), C.byref(det)))
            rows = detect["na"] * Ho * Wo
            self._track(list(srcs), [(detect["x_out"], 0, 1 << 30)] +
                        ([(detect["z_out"], detect["row_offset"], detect["row_offset"] + rows)] if detect["z_out"] is not None else []))
        else:
            check(lib().yp_plan_add_conv(self.handle, C.byref(d)))
            self._track(list(srcs) + [res] + ([extra["weight_view"]] if "weight_view" in extra else []) + ([post[3]] if post is not None else []), [out, out2])
        # algorithmic work: MAC*2 with the REAL channel counts (BASELINE.md section 2 convention)
        Kreal = Cin * R * (S * (2 if pair else 1))
        if thin:
            Kreal = (w.shape[1]) * w.shape[2] * w.shape[3]
        M = d.B * Ho * Wo
        eb = 4 if self.code == _hip.YP_F32 else 2
        in_elems = d.B * sum((v.H * v.W * v.C) for v in srcs)
        bytes_ = in_elems * eb + M * Cout * (4 if out_f32 else eb) * (2 if (detect is not None and detect['z_out'] is not None) else 1) + Cout * Kreal * eb
        flops = 2 * M * Cout * Kreal
        if pre is not None:       # the 1x1 of the fused Bottleneck: its algorithmic work (not the halo recompute) and its filter
            flops += 2 * M * Cin * Cin
            bytes_ += Cin * Cin * eb
        if post is not None:      # the C3 tail: 1x1 over 2*Cout channels; reads the other branch, writes 2*Cout channels (counted in M * out.C above? no: add)
            flops += 2 * M * (2 * Cout) * (2 * Cout)
            bytes_ += (M * Cout + M * Cout + 4 * Cout * Cout) * eb
        self.records.append(OpRecord(self.name(), "conv", flops, bytes_, M, Cout, Kreal))
        return out

    def stem(self, w, bias, act, H, W):
        """Fused stem (yp_stem_conv): returns (output view, launch(x_nchw_fp32)).  The launch is NOT part of the plan: it
        reads the caller's tensor directly, so it is enqueued eagerly right before the plan / graph replay."""
        Cout = w.shape[0]
        out = self.new_buf(H // 2, W // 2, Cout).view()
        wp, bp, Kpad, _ = pack_stem_weight(w, bias, self.code, self.device)
        self.keep += [wp, bp]
        ov = out.c()
        B, code = self.B, self.code

        def launch(x, stream=None):
            check(lib().yp_stem_conv(x.data_ptr(), B, x.shape[1], x.shape[2], x.shape[3], wp.data_ptr(), Kpad,
                                     bp.data_ptr() if bias is not None else None, act, ov, code, _hip.stream_ptr(stream)))
        Kreal = w.shape[1] * w.shape[2] * w.shape[3]
        M = B * (H // 2) * (W // 2)
        self.stem_record = OpRecord(self.name("stem"), "conv", 2 * M * Cout * Kreal, B * w.shape[1] * H * W * 4 + M * Cout * 2 + Cout * Kreal * 2, M, Cout, Kreal)
        return out, launch

    def stem_conv2(self, w1, b1, act1, H, W, w2, b2, act2, post=None):
        """Fused stem + the 3x3 / stride-2 convolution behind it (YpConvDesc.stem_*: the stem's 32-channel output stays in LDS).  Returns
        (Conv2's output view, launch(x_nchw_fp32)); like stem(), the launch is enqueued eagerly in front of the plan replay.
        post = (w3 [64, 64, 1, 1], b3, act3, out view, out2 view | None): the pointwise convolution behind Conv2 (C3.cv1 + C3.cv2 of the
        next block) runs inside the same launch and writes out / out2; Conv2's own output is never materialised (returns (None, launch))."""
        C1, C2 = w1.shape[0], w2.shape[0]
        assert C1 == 32 and tuple(w2.shape[1:]) == (32, 3, 3) and C2 <= 64 and H % 4 == 0 and W % 4 == 0
        H1, W1, H2, W2 = H // 2, W // 2, H // 4, W // 4
        if post is not None:
            w3, b3, act3, o1, o2 = post
            assert C2 == 64 and tuple(w3.shape) == (64, 64, 1, 1) and o1.C + (o2.C if o2 is not None else 0) == 64
            out = o1
        else:
            out = self.new_buf(H2, W2, round_up(C2, 8)).view()
        wp1, bp1, Kpad1, _ = pack_stem_weight(w1, b1, self.code, self.device)
        wp2, bp2, Kpad2, Npad2 = pack_conv_weight(w2, b2, self.code, self.device)
        self.keep += [wp1, bp1, wp2, bp2]
        d = YpConvDesc()
        hidden = _hip.YpView()      # describes the stem's output, which is never materialised
        hidden.ptr, hidden.H, hidden.W, hidden.cstride, hidden.coff, hidden.C, hidden.ups = wp2.data_ptr(), H1, W1, C1, 0, C1, 0
        d.in0, d.in1, d.out, d.res, d.out2 = hidden, NULL_VIEW, out.c(), NULL_VIEW, (post[4].c() if post is not None and post[4] is not None else NULL_VIEW)
        if post is not None:
            wp3, bp3, Kpad3, Npad3 = pack_conv_weight(w3, b3, self.code, self.device)
            self.keep += [wp3, bp3]
            d.post_weight, d.post_bias = wp3.data_ptr(), (bp3.data_ptr() if b3 is not None else None)
            d.post_Kpad, d.post_Npad, d.post_act = Kpad3, Npad3, act3
        d.weight, d.bias = wp2.data_ptr(), (bp2.data_ptr() if b2 is not None else None)
        d.dtype, d.out_f32, d.B = self.code, 0, self.B
        d.Hi, d.Wi, d.Ho, d.Wo = H1, W1, H2, W2
        d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w = 3, 3, 2, 2, 1, 1
        d.dil_h = d.dil_w = 1
        d.ksplit = 1
        d.Kpad, d.Npad, d.act, d.tile, d.tail_zero = Kpad2, Npad2, act2, 0, 1
        d.stem_weight, d.stem_bias = wp1.data_ptr(), (bp1.data_ptr() if b1 is not None else None)
        d.stem_Kpad, d.stem_act = Kpad1, act1
        self.keep.append(d)

        def launch(x, stream=None):
            d.stem_x, d.stem_C = x.data_ptr(), x.shape[1]
            check(lib().yp_conv2d(C.byref(d), _hip.stream_ptr(stream)))
        K1, K2 = w1.shape[1] * w1.shape[2] * w1.shape[3], 32 * 9
        M1, M2 = self.B * H1 * W1, self.B * H2 * W2
        fl3 = 2 * M2 * 64 * 64 if post is not None else 0
        self.stem_record = OpRecord(self.name("stem+Conv2" + ("+cv1+cv2" if post is not None else "")), "conv", 2 * M1 * C1 * K1 + 2 * M2 * C2 * K2 + fl3,
                                    self.B * w1.shape[1] * H * W * 4 + M2 * C2 * 2 + (C1 * K1 + C2 * K2 + (64 * 64 if post is not None else 0)) * 2, M2, C2, K2)
        return (None if post is not None else out), launch

    def op(self, code, reads, writes, name, **kw):
        """Append a generic launch record (training-path kernels); kw: v=[Views], f/g/p=[tensors|ptr], n=[sizes], i=[ints], s=[floats]."""
        a = YpOpArgs()
        a.op = code
        for j, v in enumerate(kw.get("v", [])):
            a.v[j] = v.c() if v is not None else NULL_VIEW
        for key in ("f", "g", "p"):
            arr = getattr(a, key)
            for j, t in enumerate(kw.get(key, [])):
                arr[j] = (t.data_ptr() if isinstance(t, torch.Tensor) else t) if t is not None else None
        for j, t in enumerate(kw.get("n", [])):
            a.n[j] = int(t)
        for j, t in enumerate(kw.get("i", [])):
            a.i[j] = int(t)
        for j, t in enumerate(kw.get("s", [])):
            a.s[j] = float(t)
        check(lib().yp_plan_add_op(self.handle, C.byref(a)))
        self._track(reads, writes)
        self.records.append(OpRecord(self.name(name), "aux"))

    @contextlib.contextmanager
    def side(self, enable=True):
        """Ops added inside this block go to the plan's SIDE lane (a second stream; under graph capture a parallel branch): each waits for
        every op added before it and for the side ops before it, and nothing outside the block waits for them before the plan ends.  For
        branches whose results only the caller reads (the keypoint / descriptor heads and the Detect levels of the inference plan): the
        plan then replays as a main chain + one side chain instead of per-slice dependency edges.  YP_INFER_LANES=0 disables it."""
        n0 = lib().yp_plan_num_ops(self.handle)
        yield
        if enable and sw("YP_INFER_LANES") != "0":
            for j in range(n0, lib().yp_plan_num_ops(self.handle)):
                check(lib().yp_plan_set_lane(self.handle, j, _hip.LANE_SIDE))
                self.has_lanes = True
                self.__dict__.setdefault("side_ops", set()).add(j)

    def callback(self, fn, name="callback"):
        """Append a callback op: at replay `fn(stream_ptr)` is called (stream_ptr: the raw HIP stream of the op's lane as an int) and may
        enqueue launches of its own there.  Plans with callbacks replay eagerly (ExecPlan.instantiate_graph is skipped by the callers)."""
        def tramp(user, stream):
            try:
                fn(int(stream or 0))
                return 0
            except Exception as e:               # an exception cannot cross the C frame: keep it for the caller of run()
                self.callback_error = e
                return -1
        cfn = _CALLBACK_T(tramp)
        self.keep.append(cfn)
        self.__dict__.setdefault("callbacks", []).append(cfn)
        check(lib().yp_plan_add_callback(self.handle, C.cast(cfn, C.c_void_p), None))
        self._track([], [])
        self.records.append(OpRecord(self.name(name), "aux"))
        self.has_callbacks = True

    def set_lane(self, lane):
        """Put the op added last on a schedule lane (_hip.LANE_SIDE: beside the following ops; _hip.LANE_JOIN: after all side ops)."""
        j = lib().yp_plan_num_ops(self.handle) - 1
        check(lib().yp_plan_set_lane(self.handle, j, lane))
        if lane == _hip.LANE_JOIN:
            self.__dict__.setdefault("join_ops", []).append(j)

    def refresh(self):
        """Re-derive every packed weight from its source (call before each training step)."""
        for fn in self.refreshers:
            fn()

    def _autotune(self, d, det, key, stat_group_px=None):
        """Pick the fastest kernel variant for this convolution by timing each candidate on the plan's own buffers
        (HIP events on the current stream).  Every variant computes the same convolution; the choice is cached per
        signature so that equal layers always run the same kernel within a process."""
        # Data-parallel runs: every rank builds the same plans in the same order; rank 0 times the candidates and broadcasts its choice, so that
        # all replicas run the SAME kernel variant for a layer (timings differ by a few percent from GPU to GPU: per-process tuning let ranks
        # pick different tiles -- different fp32 summation orders and step times across the replicas of one job).  YP_TUNE_SHARED=0: per rank.
        # The exchange does not depend on the state of a rank's cache: EVERY call takes part in one broadcast (rank 0 sends its choice, cached
        # or freshly timed, with a checksum of the signature), so ranks whose caches differ (an extra validation plan on rank 0, a restarted
        # rank) cannot fall out of step; a receiver checks the signature, and that the tile applies to ITS descriptor, before it adopts it.
        shared = _shared_tuning_group()
        if shared is None:
            if key in _TUNE_CACHE:
                return _TUNE_CACHE[key]
        else:
            import zlib
            sig = zlib.crc32(repr(key).encode()) & 0x7fffffff
            if shared[1] == 0:
                choice = _TUNE_CACHE[key] if key in _TUNE_CACHE else self._autotune_local(d, det, key, stat_group_px)
                msg = torch.tensor([sig, int(choice[0]), int(round((choice[1] or 0.0) * 1e9))], dtype=torch.int64, device=self.device)
                shared[0].broadcast(msg, src=0)
                return choice
            msg = torch.zeros(3, dtype=torch.int64, device=self.device)
            shared[0].broadcast(msg, src=0)
            rsig, best, ns = (int(v) for v in msg.tolist())
            ok = rsig == sig
            if ok:                                     # does rank 0's variant apply to this rank's descriptor?
                d.tile = best
                ok = (lib().yp_conv2d_detect(C.byref(d), C.byref(det), _hip.stream_ptr()) if det is not None
                      else lib().yp_conv2d(C.byref(d), _hip.stream_ptr())) == 0
            if ok:
                _TUNE_CACHE[key] = (best, (ns * 1e-9 if ns > 0 else None))
                return _TUNE_CACHE[key]
            import warnings
            warnings.warn(f"shared autotuning: rank {shared[1]} received tile {best} for another signature or one that does not apply here "
                          f"({self.name()}); the ranks are building different plans -- tuning this layer locally")
            if key in _TUNE_CACHE:
                return _TUNE_CACHE[key]
        return self._autotune_local(d, det, key, stat_group_px)

    def _autotune_local(self, d, det, key, stat_group_px=None):
        """Time every applicable variant on this GPU and cache the fastest (see _autotune)."""
        st = _hip.stream_ptr()
        if det is not None:
            run = lambda: lib().yp_conv2d_detect(C.byref(d), C.byref(det), st)
        else:
            run = lambda: lib().yp_conv2d(C.byref(d), st)
        best, best_ms = 0, None
        rnd = sw("YP_TUNE_RANDOM") or None        # stress mode (tests): a pseudo-random applicable variant per signature instead of the fastest
        applicable = []
        for cand in _TUNE_CANDIDATES:
            if det is not None and 10 <= cand <= 19:
                continue
            if stat_group_px and d.bn_partial and stat_group_px % _STAT_ROW_PX.get(cand, 64 if cand < 10 or cand > 15 else 1):
                continue                              # a statistics row of this variant would straddle two BatchNorm statistics groups
            d.tile = cand
            if run() != 0:
                continue                              # variant does not apply to this convolution
            if rnd is not None:
                applicable.append(cand)
                continue
            if best_ms is None:
                # the first timed candidate of a signature follows host-side work (descriptor set-up, filter packing) during which the GPU idled
                # and dropped its clocks: tools/conv_bench.py's first column measured 2-4 x slow for exactly this reason.  A burst of untimed
                # launches in front of it, so that every candidate is timed at the same clocks.
                for _ in range(2 * _TUNE_ITERS):
                    run()
            run()
            iters = _TUNE_ITERS
            if _TUNE_COLD:
                # COLD timing: inside a step a layer runs once, with its filters and input fetched through the fabric (the step's other
                # tensors have been through the L2s since the last time); back-to-back launches of one layer time it with a hot L2.
                # A write sweep over _TUNE_COLD MB in front of every timed launch evicts the eight L2s (4 MB each).
                global _TUNE_FLUSH
                if _TUNE_FLUSH is None or _TUNE_FLUSH.device != self.device:
                    _TUNE_FLUSH = torch.empty((_TUNE_COLD << 20,), dtype=torch.uint8, device=self.device)
                evs = []
                for it in range(iters):
                    _TUNE_FLUSH.fill_(it & 1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run()
                    e1.record()
                    evs.append((e0, e1))
                evs[-1][1].synchronize()
                ts = sorted(a.elapsed_time(b) for a, b in evs)
                ms = ts[len(ts) // 2] * 8.0                # median launch, in the cache's units of 8 launches
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) * 8.0 / iters          # (kept in units of 8 launches: the cache stores ms / 8)
            if sw("YP_TUNE_DEBUG"):
                print(f"[tune] {self.name():40s} cand {cand:2d}: {ms / 8 * 1e3:7.1f} us", flush=True)
            if best_ms is None or ms < best_ms:
                best, best_ms = cand, ms
        if rnd is not None and applicable:
            import random
            best, best_ms = random.Random(f"{rnd}:{len(_TUNE_CACHE)}").choice(applicable), None
            force = dict(tuple(int(v) for v in kv.split(":")) for kv in sw("YP_TUNE_FORCE").split(",") if kv)
            if force:                                             # explicit mixture: signature index -> variant, everything else the first one
                best = force.get(len(_TUNE_CACHE), applicable[0])
            if sw("YP_TUNE_DEBUG"):
                print(f"[tune-random] {self.name():44s} pick {best:2d} of {applicable} zs={int(d.in0_zero_stuffed)} k={d.R}x{d.S} s={d.stride_h} Cin={d.in0.C}+{d.in1.C} N={d.Npad} M={d.B * d.Ho * d.Wo}", flush=True)
        _TUNE_CACHE[key] = (best, (best_ms / 8 if best_ms is not None else None))
        return _TUNE_CACHE[key]

    def sppf_pool(self, x, y1, y2, y3):
        check(lib().yp_plan_add_sppf_pool(self.handle, x.c(), y1.c(), y2.c(), y3.c(), self.B, self.code))
        self._track([x], [y1, y2, y3])
        eb = 4 if self.code == _hip.YP_F32 else 2
        self.records.append(OpRecord(self.name("m"), "pool", 0, 4 * self.B * x.H * x.W * x.C * eb))

    def l2norm(self, src, dst, C_):
        check(lib().yp_plan_add_l2norm(self.handle, src.c(), dst.c(), self.B, C_))
        self._track([src], [dst])
        self.records.append(OpRecord(self.name("l2norm"), "l2norm", 0, 2 * self.B * src.H * src.W * C_ * 4))

    def detect_decode(self, raw, na, no, stride, anchors_px, x_out, z_out, rows_total, row_offset):
        arr = (C.c_float * (na * 2))(*[float(a) for a in anchors_px])
        check(lib().yp_plan_add_detect_decode(self.handle, raw.c(), self.B, na, no, float(stride), arr,
                                              x_out.data_ptr(), z_out.data_ptr() if z_out is not None else None,
                                              rows_total, row_offset))
        rows = na * raw.H * raw.W
        self._track([raw], [(x_out, 0, 1 << 30)] + ([(z_out, row_offset, row_offset + rows)] if z_out is not None else []))
        n = self.B * na * raw.H * raw.W * no * 4
        self.records.append(OpRecord(self.name("decode"), "decode", 0, n * (3 if z_out is not None else 2)))

    def finish(self, parallel=True):
        """Freeze the plan; attach the data dependencies used when it is captured into a hipGraph."""
        assert len(self.accesses) == len(self.records) == lib().yp_plan_num_ops(self.handle)
        if sw("YP_GRAPH_LINEAR") == "1":      # A/B: replay every plan as a linear chain
            parallel = False
        if getattr(self, "has_lanes", False):             # explicit schedule lanes: the captured topology (main chain + side chain) is the schedule
            parallel = False
            # the lane assignment must agree with the data dependencies: nothing on the main lane may touch what a side op writes or is
            # still reading (a side op waits for every op in front of it, so the other direction is covered by construction) -- unless a
            # LANE_JOIN op in between has made the main lane wait for the side lane
            side = getattr(self, "side_ops", set())
            joins = getattr(self, "join_ops", [])
            for j, d in enumerate(self.dependencies()):
                if j not in side:
                    bad = [i for i in d if i in side and not any(i < k <= j for k in joins)]
                    if bad:
                        raise _hip.YpError(f"plan lanes: main-lane op {j} ({self.records[j].name}) depends on side-lane op(s) "
                                           f"{[(i, self.records[i].name) for i in bad]}")
        self.deps = self.dependencies() if parallel else None
        if self.deps is not None:
            for j, d in enumerate(self.deps):
                arr = (C.c_int * max(len(d), 1))(*d)
                check(lib().yp_plan_set_deps(self.handle, j, arr, len(d)))
        return ExecPlan(self)


class ExecPlan:
    """A finished plan: run()/profile()/time() replay it on the current stream."""

    def __init__(self, pb):
        self.handle, self.keep, self.records = pb.handle, pb.keep, pb.records
        self.B, self.code, self.device = pb.B, pb.code, pb.device
        self.deps = pb.deps
        self.refreshers = pb.refreshers
        self.graph = False
        self.has_lanes = bool(getattr(pb, "has_lanes", False))     # ops on the side lane: replays on two streams (graph: a forked branch)
        self.has_callbacks = bool(getattr(pb, "has_callbacks", False))
        self._pb = pb if self.has_callbacks else None              # (callback trampolines report exceptions through the builder)
        self.parallel = self.has_lanes

    def num_ops(self):
        return lib().yp_plan_num_ops(self.handle)

    def instantiate_graph(self):
        """Capture the launch list into a hipGraph (needs a non-default stream for capture)."""
        if self.graph:
            return
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            check(lib().yp_plan_instantiate_graph(self.handle, _hip.stream_ptr(s)))
        torch.cuda.current_stream().wait_stream(s)
        # Everything the build enqueued (zero fills of the output tensors, filter uploads) must have retired before the FIRST replay: a graph
        # with a side branch (schedule lanes) runs that branch on a stream of its own, which the runtime orders behind the launch stream's
        # earlier KERNELS only through the graph's root -- measured: the Detect outputs of the first replay came out zero-filled in parts.
        torch.cuda.synchronize(self.device)
        self.graph = True
        self.parallel = bool(lib().yp_plan_graph_is_parallel(self.handle)) or self.has_lanes

    def refresh(self):
        for fn in self.refreshers:
            fn()

    def run(self, stream=None):
        rc = lib().yp_plan_run(self.handle, _hip.stream_ptr(stream))
        if rc != 0 and self._pb is not None and getattr(self._pb, "callback_error", None) is not None:
            e, self._pb.callback_error = self._pb.callback_error, None
            raise e
        check(rc)

    def profile(self, stream=None):
        n = self.num_ops()
        ms = (C.c_float * n)()
        check(lib().yp_plan_profile(self.handle, _hip.stream_ptr(stream), ms))
        return list(ms)

    def time(self, iters, stream=None):
        ms = C.c_float()
        check(lib().yp_plan_time(self.handle, _hip.stream_ptr(stream), iters, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.handle:
                lib().yp_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# eager helpers used around a plan
# ---------------------------------------------------------------------------------------------
def pack_input(x, view, code, stream=None):
    """NCHW fp32 cuda tensor -> NHWC view (channels zero padded)."""
    B, C_, H, W = x.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    check(lib().yp_pack_input(x.data_ptr(), B, C_, H, W, view.c(), code, _hip.stream_ptr(stream)))


def unpack_nchw(view, src_code, B, C_, stream=None):
    out = torch.empty((B, C_, view.H, view.W), dtype=torch.float32, device=view.buf.t.device)
    check(lib().yp_unpack_nchw(view.c(), src_code, B, C_, out.data_ptr(), _hip.stream_ptr(stream)))
    return out
