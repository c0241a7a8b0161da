"""Building blocks with the reference's names, constructor signatures and state_dict layout
(reference: src/models/common.py:12-34 Conv, :79-89 Bottleneck, :123-135 C3, :213-229 SPPF).

torch.nn.Conv2d / BatchNorm2d instances are kept purely as parameter containers so that
checkpoints are key- and shape-compatible; their forward() is never called.  Each block knows
how to `emit` itself into a PlanBuilder; `forward(x)` runs a cached single-block plan so blocks
can be called (and tested) on their own like the reference's.
"""
import os
from ..switches import sw

import torch
import torch.nn as nn

from .. import _hip
from ..plan import PlanBuilder, View, pack_input, unpack_nchw, round_up


def autopad(k, p=None):
    """'same' padding for odd kernels (reference: models/common.py:12-16)."""
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


bn_params = {'eps': 1e-3, 'momentum': 0.03}     # reference: models/common.py:18-20


def fold_bn(conv, bn):
    """Inference-time BN folding, same algebra as utils/torch_utils_yolo.py:194-214.
    Returns (w[O,I,R,S], b[O]) fp32."""
    w = conv.weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    wf = w * scale.view(-1, 1, 1, 1)
    b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(scale)
    bf = b0 * scale + bn.bias.detach().float() - bn.weight.detach().float() * bn.running_mean.detach().float() / torch.sqrt(
        bn.running_var.detach().float() + bn.eps)
    return wf, bf


_GENERATION = [0]


def weights_generation():
    """Number of optimizer steps taken in this process (any optimizer) + explicit invalidations: see HipModule._weights_version."""
    return _GENERATION[0]


def invalidate_packed_weights():
    """Tell every plan that the fp32 master parameters may have changed behind PyTorch's version counters: updates made through
    `.data` (p.data.copy_ / add_, dist.broadcast(p.data), hand-written SGD or EMA) bump neither Tensor._version nor the
    optimizer-step count.  Cheap (a counter); the next forward re-derives the packed 16-bit filters.  Called by
    dp.GradAllReducer.broadcast_parameters, Model.load_state_dict and ModelEMA; call it after any other `.data` write."""
    _GENERATION[0] += 1


def _count_optimizer_step(*_args, **_kwargs):
    _GENERATION[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_count_optimizer_step)


class HipModule(nn.Module):
    """Base class: standalone forward through a cached native plan."""

    compute_dtype = "f16"
    # True only for a model class whose training buffers are read by plan ops alone (training.TrainGraph._drop_unread_16bit_copies may then stop
    # writing 16-bit copies nobody reads in fp8 mode).  NOT inherited by intent: a subclass that adds torch-side differentiation must not get it.
    plans_cover_all_reads = False
    _NATIVE_CACHES = ("_plans", "_train_graphs", "_pack_states", "_mods_cache", "_frozen_version")      # per-object native plans / graphs / caches: never copied or pickled

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._NATIVE_CACHES:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._NATIVE_CACHES}

    def _weights_version(self):
        """Changes whenever the parameters / buffers may have changed: tensor version counters and storage addresses, plus the
        process-wide optimizer-step count -- fused optimizers (torch.optim.Adam(fused=True)) update parameters WITHOUT bumping
        Tensor._version, so a key built on the counters alone keeps replaying plans packed from the old filters."""
        # (Module.parameters() / buffers() re-walk the module tree with de-duplication on every call: 1.5 ms per forward for YOLOPoint-l,
        # as much host time as the forward takes on the device.  The module list is cached and re-walked only when the tree changed --
        # detected by the IDENTITIES of every module's children (fuse() / add_module / `net.ConvDet = new_head` all change them; a child
        # count alone misses a same-shape replacement, which would keep replaying a plan packed from the old module's filters) -- or an
        # explicit invalidation; the tensors are read from the modules' own dicts every time, so replaced parameters are seen.)
        gen = weights_generation()
        fz = self.__dict__.get("_frozen_version")
        if fz is not None and fz[0] == gen:               # freeze_weights(): the caller vouches that nothing but optimizers / load_state_dict
            return fz                                      # (which bump the generation) touches the parameters -- no walk per forward
        cache = self.__dict__.get("_mods_cache")
        if cache is None or cache[0] != gen or cache[1] != tuple(id(c) for m in cache[2] for c in m._modules.values()):
            mods = list(self.modules())
            cache = self.__dict__["_mods_cache"] = (gen, tuple(id(c) for m in mods for c in m._modules.values()), mods)
        return (gen,) + tuple((t._version, t.data_ptr()) for m in cache[2] for d in (m._parameters, m._buffers) for t in d.values() if t is not None)

    def freeze_weights(self, frozen=True):
        """Inference servers (frontend.YoloPointFrontend) load a checkpoint once: with frozen weights the per-forward walk over every
        parameter's version counter (0.3 ms of host time for YOLOPoint-l, in front of the first launch of every frame) is skipped.  An
        optimizer step, load_state_dict or invalidate_packed_weights() still invalidates the plans (they bump the generation); in-place
        edits of a parameter tensor are NOT seen until freeze_weights(False)."""
        self.__dict__.pop("_frozen_version", None)
        if frozen:
            self.__dict__["_frozen_version"] = self._weights_version()
        return self

    def _plan_key(self, x):
        return (tuple(x.shape), _hip.dtype_code(self.compute_dtype), self.training, self._weights_version(), x.device.index)

    def _standalone_out_channels(self):
        raise NotImplementedError

    @_hip.guarded
    def forward(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise _hip.YpError(f"{type(self).__name__}.forward needs a cuda (HIP) tensor: the hot path has no CPU fallback")
        if self.training:
            raise _hip.YpError("train-mode (batch-statistics) forward is not part of this build yet; call .eval()")
        x = x.contiguous().float()
        key = self._plan_key(x)
        cache = self.__dict__.setdefault("_plans", {})
        if key not in cache:
            cache.clear()
            B, C_, H, W = x.shape
            code = _hip.dtype_code(self.compute_dtype)
            pb = PlanBuilder(B, code, x.device)
            ce = pb.ce
            inb = pb.new_buf(H, W, 4 if C_ <= 4 else round_up(C_, ce))   # <=4 channels: image-like (stem) input
            out = self.emit(pb, inb.view())
            cache[key] = (pb.finish(), inb, out)
        plan, inb, out = cache[key]
        pack_input(x, inb.view(), plan.code)
        plan.run()
        return unpack_nchw(out, plan.code, x.shape[0], self._standalone_out_channels())


class Conv(HipModule):
    """Conv2d(bias=False) + BatchNorm2d + SiLU  (reference: models/common.py:22-34)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        if g != 1:
            raise _hip.YpError("grouped convolutions are not on the YOLOPoint hot path")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, **bn_params)
        self.act = nn.SiLU(inplace=True) if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def _standalone_out_channels(self):
        return self.conv.out_channels

    def folded(self):
        if hasattr(self, "bn"):
            return fold_bn(self.conv, self.bn)
        return self.conv.weight.detach().float(), (self.conv.bias.detach().float() if self.conv.bias is not None else None)

    def emit(self, pb, x, out=None, res=None, out_f32=False):
        w, b = self.folded()
        act = _hip.YP_ACT_SILU if isinstance(self.act, nn.SiLU) else _hip.YP_ACT_NONE
        if not isinstance(self.act, (nn.SiLU, nn.Identity)):
            raise _hip.YpError(f"unsupported activation {type(self.act).__name__}")
        k, s, p = self.conv.kernel_size[0], self.conv.stride[0], self.conv.padding[0]
        pb.scope.append("conv")
        try:
            return pb.conv(x, w, b, k, s, p, act, out=out, res=res, out_f32=out_f32)
        finally:
            pb.scope.pop()


class Bottleneck(HipModule):
    """x + cv2(cv1(x))  (reference: models/common.py:79-89); the add is fused into cv2's epilogue."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def _standalone_out_channels(self):
        return self.cv2.conv.out_channels

    # True / False / "auto" (time both forms when the plan is built): run cv1 as the LDS prologue of cv2's 3x3 kernel
    # (one launch, hidden tensor never in HBM).  YP_FUSE_BOTTLENECK=0|1 overrides for A/B measurements.
    # Default True: back-to-back timing of the two-launch form ("auto") under-estimates its in-chain latency; the fused
    # form measured +5% whole-net throughput on YOLOPoint-s (bs8 640x640 f16) vs +3.8% for "auto".
    @property
    def fuse(self):
        return {"0": False, "auto": "auto"}.get(sw("YP_FUSE_BOTTLENECK"), True)       # (read when a plan is emitted)

    def _fusable(self, pb, x):
        c1, c_, c2 = self.cv1.conv.in_channels, self.cv1.conv.out_channels, self.cv2.conv.out_channels
        allowed = (32, 64, 128)
        return (self.fuse and isinstance(x, View) and x.ups == 0 and c1 == c_ == c2 and c_ in allowed and pb.code != _hip.YP_F32
                and self.cv2.conv.kernel_size == (3, 3) and self.cv2.conv.stride == (1, 1) and self.cv2.conv.padding == (1, 1)
                and isinstance(self.cv1.act, nn.SiLU) and isinstance(self.cv2.act, nn.SiLU))

    def emit(self, pb, x, out=None):
        res = x if self.add else None
        if self._fusable(pb, x):
            (w1, b1), (w2, b2) = self.cv1.folded(), self.cv2.folded()
            if out is None:
                out = pb.new_buf(x.H, x.W, w2.shape[0]).view()
            pre = {"pre": (w1, b1, _hip.YP_ACT_SILU)}
            fused = True
            if self.fuse == "auto" and pb.autotune:
                t = pb.new_buf(x.H, x.W, w1.shape[0]).view()
                ms_f = pb.conv(x, w2, b2, 3, 1, 1, _hip.YP_ACT_SILU, out=out, res=res, extra=dict(pre, dry_run=True))
                ms_1 = pb.conv(x, w1, b1, 1, 1, 0, _hip.YP_ACT_SILU, out=t, extra={"dry_run": True})
                ms_2 = pb.conv(t, w2, b2, 3, 1, 1, _hip.YP_ACT_SILU, out=out, res=res, extra={"dry_run": True})
                fused = ms_f is not None and (ms_1 is None or ms_2 is None or ms_f <= ms_1 + ms_2)
            if fused:
                pb.scope.append("cv1>cv2")
                try:
                    return pb.conv(x, w2, b2, 3, 1, 1, _hip.YP_ACT_SILU, out=out, res=res, extra=pre)
                finally:
                    pb.scope.pop()
        pb.scope.append("cv1"); t = self.cv1.emit(pb, x); pb.scope.pop()
        pb.scope.append("cv2"); y = self.cv2.emit(pb, t, out=out, res=res); pb.scope.pop()
        return y


class C3(HipModule):
    """cv3(cat(m(cv1(x)), cv2(x)))  (reference: models/common.py:123-135).
    The concat is a shared buffer: the last Bottleneck writes channels [0,c_), cv2 writes [c_,2c_)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)))

    # YP_FUSE_C3_TAIL=0 disables the cv3-in-the-last-Bottleneck fusion (A/B measurements)
    fuse_tail = sw("YP_FUSE_C3_TAIL") != "0"

    def _standalone_out_channels(self):
        return self.cv3.conv.out_channels

    def merged_cv12(self):
        """cv1 and cv2 as ONE 1x1 filter [2 c_, c1] + bias (BN folded): they read the same input."""
        (w1, b1), (w2, b2) = self.cv1.folded(), self.cv2.folded()
        return torch.cat((w1, w2), 0), torch.cat((b1, b2), 0)

    def emit(self, pb, x, out=None, pre=None):
        """pre = (t, cat): cv1 + cv2 have already been emitted by the caller (inside the launch that produces this block's input:
        PlanBuilder.stem_conv2(post=...)) into t = cv1's output view and channels [c_, 2 c_) of the concat buffer `cat`."""
        c_ = self.cv1.conv.out_channels
        if pre is not None:
            t, cat = pre
            x0 = t
        else:
            x0 = x[0] if isinstance(x, (list, tuple)) else x
            cat = pb.new_buf(x0.LH, x0.LW, 2 * c_)
            # cv1 and cv2 are 1x1 convolutions of the same input: one launch, N = 2*c_, split destination
            # (cv1 -> its own buffer feeding the bottleneck chain, cv2 -> channels [c_, 2c_) of the concat)
            w12, b12 = self.merged_cv12()
            t = pb.new_buf(x0.LH, x0.LW, c_).view()
            pb.scope.append("cv1+cv2")
            pb.conv(x, w12, b12, 1, 1, 0, _hip.YP_ACT_SILU, out=t, out2=cat.view(c_, c_))
            pb.scope.pop()
        n = len(self.m)
        last = self.m[n - 1]
        # C3 tail fusion: cv3 runs inside the last Bottleneck's kernel (its output never reaches HBM); hidden widths 32 / 64
        fuse_tail = (self.fuse_tail and c_ in (32, 64) and last.fuse is True and last._fusable(pb, t) and isinstance(self.cv3.act, nn.SiLU)
                     and self.cv3.conv.out_channels == 2 * c_)
        for i, blk in enumerate(self.m):
            pb.scope.append(f"m.{i}")
            if i == n - 1 and fuse_tail:
                (wa, ba), (wb, bb), (w3, b3) = blk.cv1.folded(), blk.cv2.folded(), self.cv3.folded()
                if out is None:
                    out = pb.new_buf(x0.LH, x0.LW, 2 * c_).view()
                pb.scope.append("cv1>cv2>cv3")
                y = pb.conv(t, wb, bb, 3, 1, 1, _hip.YP_ACT_SILU, out=out, res=t if blk.add else None,
                            extra={"pre": (wa, ba, _hip.YP_ACT_SILU), "post": (w3, b3, _hip.YP_ACT_SILU, cat.view(c_, c_))})
                pb.scope.pop(); pb.scope.pop()
                return y
            t = blk.emit(pb, t, out=cat.view(0, c_) if i == n - 1 else None)
            pb.scope.pop()
        pb.scope.append("cv3"); y = self.cv3.emit(pb, cat.view(), out=out); pb.scope.pop()
        return y


class Bottleneckv8(HipModule):
    """cv2(cv1(x)) [+ x], both k x k  (reference: models/common.py:91-103)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def _standalone_out_channels(self):
        return self.cv2.conv.out_channels

    def emit(self, pb, x, out=None):
        pb.scope.append("cv1"); t = self.cv1.emit(pb, x); pb.scope.pop()
        pb.scope.append("cv2"); y = self.cv2.emit(pb, t, out=out, res=x if self.add else None); pb.scope.pop()
        return y


class C2f(HipModule):
    """cv2(cat(chunk(cv1(x), 2) + [m_i(previous)]))  (reference: models/common.py:151-171).  cv1 writes channels
    [0, 2c) of the concat buffer, bottleneck i reads slice (1+i) and writes slice (2+i): no chunk / cat copies."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneckv8(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def _standalone_out_channels(self):
        return self.cv2.conv.out_channels

    def emit(self, pb, x, out=None, out_f32=False, pre=None):
        """pre = the concat buffer whose channels [0, 2c) the caller's launch has already filled with cv1's output (PlanBuilder.stem_conv2(post=...))."""
        c, n = self.c, len(self.m)
        if c % 8:
            raise _hip.YpError(f"C2f hidden width {c} must be a multiple of 8")
        if pre is not None:
            cat = pre
        else:
            x0 = x[0] if isinstance(x, (list, tuple)) else x
            cat = pb.new_buf(x0.LH, x0.LW, (2 + n) * c)
            pb.scope.append("cv1"); self.cv1.emit(pb, x, out=cat.view(0, 2 * c)); pb.scope.pop()
        for i, blk in enumerate(self.m):
            pb.scope.append(f"m.{i}"); blk.emit(pb, cat.view((1 + i) * c, c), out=cat.view((2 + i) * c, c)); pb.scope.pop()
        pb.scope.append("cv2"); y = self.cv2.emit(pb, cat.view(), out=out, out_f32=out_f32); pb.scope.pop()
        return y


class SPPF(HipModule):
    """cv2(cat(x, m(x), m(m(x)), m(m(m(x)))))  (reference: models/common.py:213-229)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise _hip.YpError("SPPF pooling kernel is specialised for k=5")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def _standalone_out_channels(self):
        return self.cv2.conv.out_channels

    def emit(self, pb, x, out=None):
        c_ = self.cv1.conv.out_channels
        cat = pb.new_buf(x.LH, x.LW, 4 * c_)
        pb.scope.append("cv1"); self.cv1.emit(pb, x, out=cat.view(0, c_)); pb.scope.pop()
        pb.sppf_pool(cat.view(0, c_), cat.view(c_, c_), cat.view(2 * c_, c_), cat.view(3 * c_, c_))
        pb.scope.append("cv2"); y = self.cv2.emit(pb, cat.view(), out=out); pb.scope.pop()
        return y
