"""Training path of the YOLOPoint hot path: train-mode forward (batch-statistics BatchNorm) and the
backward pass, both as static native plans, exposed to PyTorch autograd through one Function.

Reference: src/train.py:208-252 — `model(img)`, `model(img_warp)`, losses in PyTorch, `backward()`,
`optimizer.step()`.  The losses stay PyTorch autograd; everything between the input
image and the three head outputs runs here:

  forward   conv (implicit-GEMM kernels, no bias) -> yp_bn_stats -> yp_bn_act_apply (+ residual)
  backward  yp_bn_act_bwd -> dgrad = the SAME conv kernels with the flipped / channel-transposed
            filter (stride 2: zero-stuffed input view) -> wgrad = a convolution too: the layer input and
            the output gradient are copied batch<->channel transposed ([C][H][W][B], yp_to_chwb); the
            gradient copy then plays the filter ([Cout][Ho*Wo*B]) of a dilated convolution over the input
            copy whose k x k "output pixels" are the filter taps; split-K + fp32 atomics fill the chip.

A TrainGraph owns every buffer of one forward/backward pair; two forwards of the same step (image
+ warped image) use two graphs from a small pool.  Packed weights are re-derived from the fp32
master parameters before every forward (`refresh`).

Pair mode (TrainGraph(pair=True), YOLOPoint.forward_pair): the two forwards of a training step run as ONE launch list over 2B samples
-- every convolution sees twice the pixels, every BatchNorm pass normalises the two sample sets with their own batch statistics
("statistics groups": yp_bn_*_grouped; the running statistics take the image pass's update first, then the warped pass's, as two module
calls do).  The backward is two plans: the Detect / PAN / YOLO-encoder layers over the image pass's B samples (the warped pass has no
object loss: reference train.py:220-241), then the shared trunk + keypoint / descriptor heads over all 2B samples.  Half the launches of
the two-graph schedule for the same arithmetic.
"""
import ctypes as C
import contextlib
import os
import weakref
from .switches import sw

import torch

from . import _hip
from ._hip import lib, check
from .plan import PlanBuilder, Buf, View, MasterWeight, Fp8State, round_up, pack_input, NULL_VIEW
from .models.common import weights_generation


class TrainGraph:
    def __init__(self, net, B, H, W, code, device, pair=False, fp8=False):
        """B: samples per forward pass.  pair: this graph runs two passes (image, warped image) as one launch list over 2B samples.
        fp8 (BASELINE configs[4], with code = bf16): the Conv layers whose channel counts are multiples of 64 multiply 8-bit operands --
        forward: e4m3 activations x e4m3 filters, dgrad: e5m2 output gradients x e4m3 filters, per-tensor delayed scaling (plan.Fp8State);
        results, BatchNorm, the weight gradients and everything else stay bf16 / fp32."""
        if H % 64 or W % 64:
            raise _hip.YpError("training needs image sizes that are multiples of 64")
        self.G, self.Bs = (2 if pair else 1), B          # statistics groups (= passes per launch list), samples per pass
        B = B * self.G                                   # samples per launch
        self.bG = self.G                                 # statistics groups of the backward plan being emitted
        self.net, self.B, self.H, self.W, self.code, self.device = net, B, H, W, code, device
        self.tdtype = _hip.torch_dtype(code)
        self.fwd = PlanBuilder(B, code, device)
        self.bwd = PlanBuilder(B, code, device)
        # packed (16-bit / transposed) copies of the master filters are shared by every graph over these parameters and
        # re-derived by ONE launch list, replayed once per optimizer step (not per forward / backward)
        pkey = (code, torch.device(device).index, tuple(p_.data_ptr() for p_ in net.parameters()))
        self.pack = net.__dict__.setdefault("_pack_states", {}).setdefault(pkey, {"pb": PlanBuilder(B, code, device), "cache": {}, "version": None})
        self.fwd.pack_target = (self.pack["pb"], self.pack["cache"])
        self.fp8 = bool(fp8) and code == _hip.YP_BF16
        if fp8 and not self.fp8:
            raise _hip.YpError("fp8 training runs on top of the bf16 compute path (model.bfloat16())")
        if self.fp8:
            self.pack.setdefault("fp8", Fp8State(device))
            self.fwd.fp8 = self.pack["pb"].fp8 = self.pack["fp8"]
        self.twins = {}            # fp8 mode: (address of a 16-bit activation buffer, format) -> its 1-byte twin Buf
        self.n_q8 = 0              # convolutions emitted with 8-bit operands
        self.n_q8w = 0             # weight gradients emitted with 8-bit operands
        self.twin_only = []        # (plan builder, op index, view slot, 16-bit view) of the BatchNorm outputs that also have a 1-byte twin
        self.builders = []         # every plan builder of this graph (the access lists behind _drop_unread_16bit_copies)
        self.n_twin_only = 0
        self.external_reads = []   # views that code OUTSIDE the plans reads (torch-side differentiation, probes, exporters): register_external_read()
        self.dropped_views = []    # the 16-bit views _drop_unread_16bit_copies stopped writing (NaN-filled): assert_readable() refuses them
        self.tape = []             # (branch, emitter): 'kp' feeds the keypoint / descriptor heads, 'yolo' only the Detect head
        self.branch = "kp"
        self.tape_tag = None       # 'kph': entries of the keypoint head (pair mode: their backward runs beside the YOLO-branch plan)
        self.touched = set()       # parameters whose gradient the plan being emitted writes
        self.gbufs = {}            # data_ptr of an activation buffer -> its gradient Buf
        self.gwritten = {}         # data_ptr -> list of written (lo, hi) channel ranges
        self.collect = []          # callables run after the backward plan: native buffers -> parameter gradients
        self.pgrads = {}           # parameter -> fp32 gradient tensor (reference layout)
        self.vparts = {}           # id(merged view of adjacent parameters) -> [the parameters]
        self.keep = []
        self.busy = False
        wsb = lib().yp_bn_workspace_bytes(B, H // 2, W // 2, 1024) + 8 * self.G * 2048 + 4096
        # one statistics workspace per schedule lane (ops of the two lanes run side by side)
        self._ws = {False: torch.empty(wsb, dtype=torch.uint8, device=device), True: torch.empty(wsb, dtype=torch.uint8, device=device)}
        self._side_emit = False
        self.Bpad = round_up(B, 8)
        self._nbt = None
        # device scalars the backward plans read: [0] = the upstream factor of the Detect-level gradients (1 unless a caller that hands the
        # plan UNSCALED object-loss gradients sets it: engine.TrainStep's native loss stage)
        self.head_scale = torch.ones(4, dtype=torch.float32, device=device)
        self._head_scale_val = 1.0
        self.pre_forward, self.post_forward = [], []      # host callables around every forward (padded BN parameter copies)
        # every per-layer weight-gradient accumulator (fp32 [Cin][k][k][Cout_pad]) lives in one arena that the backward plan
        # clears with a single memset
        # YP_TRAIN_LANES=1: weight-gradient kernels on a second lane of the backward graph (yp_plan_set_lane), beside the dgrad /
        # BatchNorm chain that never reads them.  Measured: 20.5 ms per step against 19.2 ms on one lane -- the concurrent kernels
        # fight over CUs / LDS / L2 (as the sub-batch stream experiment of the forward did) -- so it stays off.
        self.lanes = sw("YP_TRAIN_LANES") == "1"
        # (one launch per filter class and backward pass; per-layer launches only with the weight-gradient lane)
        self.group_wgrad = not self.lanes
        self.dw_arena = torch.zeros(round_up(2 * sum(p_.numel() for p_ in net.parameters()) + (1 << 20), 64), dtype=torch.float32, device=device)
        self.dw_used = 0
        # (per-slice slabs + ordered fold: bit-reproducible; the per-layer launches of the lane form use fp32 atomics)
        self.det_wgrad = self.group_wgrad
        self.wpart = None
        self._build()

    # ------------------------------------------------------------------ helpers
    @property
    def ws(self):
        return self._ws[self._side_emit]

    @contextlib.contextmanager
    def side_lane(self, pb, enable=True):
        """Emit the block's ops on `pb`'s side lane (PlanBuilder.side) with the side lane's own statistics workspace."""
        if not enable:
            yield
            return
        self._side_emit = True
        try:
            with pb.side():
                yield
        finally:
            self._side_emit = False

    def pgrad(self, param):
        parts = self.vparts.get(id(param))
        if parts is not None:
            # the merged tensor of adjacent parameters (c3: cv1 + cv2 as one layer): one gradient tensor, the real parameters' are its slices
            if param not in self.pgrads:
                g = torch.zeros_like(param, dtype=torch.float32)
                self.pgrads[param] = g
                off = 0
                for rp in parts:
                    self.pgrads[rp] = g.view(-1)[off:off + rp.numel()].view_as(rp)
                    off += rp.numel()
            self.touched.update(parts)
            return self.pgrads[param]
        if param not in self.pgrads:
            self.pgrads[param] = torch.zeros_like(param, dtype=torch.float32)
        self.touched.add(param)
        return self.pgrads[param]

    def gview(self, v):
        """Gradient view matching activation view `v` (same slice); returns (view, accumulate?) and marks it written."""
        key = v.buf.t.data_ptr()
        if key not in self.gbufs:
            gb = Buf(v.buf.B, v.buf.H, v.buf.W, v.buf.C, v.buf.t.dtype, self.device)
            self.gbufs[key] = gb
            self.keep.append(gb.flat)
            self.gwritten[key] = []
        if self.G > 1 and self.bG == 1 and key in self.kp_ptrs and not self.gwritten[key]:
            # pair mode, first write by the image-pass-only backward plan into the gradient of a trunk tensor (the backbone output the YOLO
            # encoder and the PAN read): the warped pass's half receives nothing from this plan -- zero it; the 2B-sample plan accumulates
            gb = self.gbufs[key]
            half = self.Bs * gb.H * gb.W * gb.C
            esz = gb.t.element_size()
            self.bwd.op(_hip.OP_MEMSET0, [], [self.T(gb.flat)], "zero_warp_half", p=[gb.flat.data_ptr() + half * esz], n=[half * esz])
        lo, hi = v.coff, v.coff + v.C
        acc = any(a < hi and lo < b for a, b in self.gwritten[key])
        self.gwritten[key].append((lo, hi))
        return View(self.gbufs[key], v.coff, v.C, 0), acc

    def gread(self, v):
        """Gradient view of `v` for reading (must have been written by the consumers' backward)."""
        key = v.buf.t.data_ptr()
        assert key in self.gbufs and any(a <= v.coff and v.coff + v.C <= b for a, b in self._merged(key)), "gradient read before write"
        return View(self.gbufs[key], v.coff, v.C, 0)

    def _merged(self, key):
        rs = sorted(self.gwritten[key])
        out = []
        for a, b in rs:
            if out and a <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], b))
            else:
                out.append((a, b))
        return out

    def T(self, t):
        return (t, 0, 1 << 30)

    # ------------------------------------------------------------------ fp8 mode
    @staticmethod
    def q8_ok(views):
        """8-bit operands need the kernels' 64-byte k chunks: every source slice a multiple of 64 channels, 16-byte aligned."""
        return all(v.geom is None and v.C % 64 == 0 and v.coff % 16 == 0 and v.cstride % 16 == 0 for v in views)

    def q8_twin(self, pb, v, fmt):
        """The 1-byte twin Buf of view `v`'s buffer and its scale slot (one twin, one scale per activation / gradient buffer and format)."""
        key = (v.buf.t.data_ptr(), fmt)
        if key not in self.twins:
            tb = Buf(v.buf.B, v.buf.H, v.buf.W, v.buf.C, torch.uint8, self.device)
            self.keep.append(tb.flat)
            self.twins[key] = tb
        return self.twins[key], pb.fp8.slot(("act", key), fmt), key

    @staticmethod
    def q8_fusable(v):
        """The BatchNorm passes write the twin themselves when their fast path serves the channel count (C/8 a power of two <= 256)."""
        ch = v.C // 8
        return v.C % 8 == 0 and 1 <= ch <= 256 and (ch & (ch - 1)) == 0

    def q8_produced(self, pb, v, fmt):
        """Arguments that make a BatchNorm pass write view `v`'s twin (forward output: fmt 0 = e4m3; dx: fmt 1 = e5m2), or None."""
        if not (self.fp8 and self.q8_ok([v]) and self.q8_fusable(v)):
            return None
        tb, slot, key = self.q8_twin(pb, v, fmt)
        pb.__dict__.setdefault("quantized", {}).setdefault(key, []).append((v.coff, v.coff + v.C))
        return View(tb, v.coff, v.C), pb.fp8.scale_ptr(slot), pb.fp8.amax_ptr(slot)

    def q8_sources(self, pb, srcs, fmt):
        """1-byte twins of the 16-bit views `srcs`, the quantisation launches that fill them where no producer did, and the scale slot
        they share.  A single source reads its buffer's own twin (reused by every consumer); channel-concatenated sources of one
        convolution need ONE scale, so they get twins private to that convolution."""
        st = pb.fp8
        if len(srcs) == 1:
            v = srcs[0]
            tb, slot, key = self.q8_twin(pb, v, fmt)
            done = pb.__dict__.setdefault("quantized", {}).setdefault(key, [])
            lo, covered = v.coff, False
            for a_, b_ in sorted(done):               # is [coff, coff + C) covered by the slices written so far?
                if a_ <= lo < b_:
                    lo = b_
            covered = lo >= v.coff + v.C
            if not covered:
                done.append((v.coff, v.coff + v.C))
                sv, dv = View(v.buf, v.coff, v.C), View(tb, v.coff, v.C)
                pb.op(_hip.OP_QUANT_FP8, [sv], [dv], "quant", v=[sv, dv], i=[self.code, pb.B, fmt], p=[st.scale_ptr(slot), st.amax_ptr(slot)])
            return [View(tb, v.coff, v.C, v.ups)], slot
        slot = st.slot(("cat", pb.handle.value, len(pb.records)), fmt)
        out = []
        for v in srcs:
            tb = Buf(pb.B, v.buf.H, v.buf.W, v.C, torch.uint8, self.device)
            self.keep.append(tb.flat)
            sv, dv = View(v.buf, v.coff, v.C), tb.view()
            pb.op(_hip.OP_QUANT_FP8, [sv], [dv], "quant", v=[sv, dv], i=[self.code, pb.B, fmt], p=[st.scale_ptr(slot), st.amax_ptr(slot)])
            out.append(View(tb, 0, v.C, v.ups))
        return out, slot

    # ------------------------------------------------------------------ forward emitters (+ tape)
    def conv_bn_act(self, m, x, out=None, res=None):
        srcs = list(x) if isinstance(x, (list, tuple)) else [x]
        conv, bn = m.conv, m.bn
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        act = _hip.YP_ACT_SILU if isinstance(m.act, torch.nn.SiLU) else _hip.YP_ACT_NONE
        f, B, code, G = self.fwd, self.B, self.code, self.G
        image = len(srcs) == 1 and srcs[0].geom is None and srcs[0].cstride == 4 and srcs[0].C == 4      # the stem: its own packed filter form
        wsrc = MasterWeight(conv.weight, mode="image" if image else 0)
        # The BatchNorm column sums come out of the convolution's epilogue (YpConvDesc.bn_partial: one partial row per block of 64 pixels in
        # the generic kernel, per pixel tile in the 3x3 halo kernels) -- no separate reduction pass over the raw output.
        chunk = 16 if code == _hip.YP_F32 else 32          # (the epilogue variant exists for the kernels' fast addressing mode: 64-byte k chunks)
        fuse_stats = (not image and all(v.C % chunk == 0 for v in srcs) and (k == 1 and s == 1 or (k == 3 and code != _hip.YP_F32)))
        partial = None
        Ho_, Wo_ = (srcs[0].LH + 2 * p - k) // s + 1, (srcs[0].LW + 2 * p - k) // s + 1
        if G > 1 and (self.Bs * Ho_ * Wo_) % 64:
            fuse_stats = False       # (a 64-pixel row block of the generic kernel would straddle the two statistics groups)
        if fuse_stats:
            rows = max(-(-(B * Ho_ * Wo_) // 64), B * -(-Ho_ // 4) * -(-Wo_ // 16) if k == 3 else 0)
            partial = torch.zeros((rows, 2, round_up(conv.out_channels, 8)), dtype=torch.float32, device=self.device)
        q8 = self.fp8 and not image and self.q8_ok(srcs) and sw("YP_FP8_FWD") != "0"
        conv_srcs, extra = srcs, (dict(bn_partial=partial, stat_group_px=(self.Bs * Ho_ * Wo_ if G > 1 else None)) if fuse_stats else {})
        x8 = None
        if q8:
            conv_srcs, slot = self.q8_sources(f, srcs, 0)
            wsrc = MasterWeight(conv.weight, q8=True)
            extra["q8"] = dict(dtype=_hip.YP_FP8, slot=slot)
            self.n_q8 += 1
            x8 = (conv_srcs, slot)           # the e4m3 twins of the sources + their scale slot: the weight gradient's operand as well
        raw = f.conv(conv_srcs, wsrc, None, k, s, p, _hip.YP_ACT_NONE, extra=extra or None)
        if fuse_stats:
            partial.zero_()          # (the autotuner ran several kernel variants, which write different row sets: start from zeros with the chosen one)
        Cc, Cp = conv.out_channels, raw.C
        mean, invstd = f.new_tensor((G * Cp,)), f.new_tensor((G * Cp,))          # [statistics group][channel]
        gamma, beta, rmean, rvar = bn.weight, bn.bias, bn.running_mean, bn.running_var
        padded = Cp != Cc
        if padded:
            # the kernels walk the padded channel count (65 -> 72 for the v52 keypoint head): give them padded parameter /
            # statistics tensors, synchronised with the module's around every pass (the padded channels are all-zero activations)
            gamma, beta, rmean = (f.new_tensor((Cp,)) for _ in range(3))
            rvar = torch.ones((Cp,), dtype=torch.float32, device=self.device)
            f.keep.append(rvar)

            def sync_in(gamma=gamma, beta=beta, rmean=rmean, rvar=rvar):
                gamma[:Cc].copy_(bn.weight.detach()); beta[:Cc].copy_(bn.bias.detach())
                rmean[:Cc].copy_(bn.running_mean); rvar[:Cc].copy_(bn.running_var)

            def sync_out(rmean=rmean, rvar=rvar):
                bn.running_mean.copy_(rmean[:Cc]); bn.running_var.copy_(rvar[:Cc])
            self.pre_forward.append(sync_in)
            self.post_forward.append(sync_out)
        if fuse_stats:
            assert partial.shape[2] == Cp
            assert f.last_bn_rows % G == 0
            f.op(_hip.OP_BN_STATS, [raw, self.T(partial)], [self.T(mean), self.T(invstd), self.T(self.ws)], "bn_stats", v=[raw], i=[code, B, f.last_bn_rows, G],
                 s=[bn.eps, bn.momentum], g=[mean, invstd, rmean, rvar], p=[self.ws, partial], n=[self.ws.numel()])
        else:
            f.op(_hip.OP_BN_STATS, [raw], [self.T(mean), self.T(invstd), self.T(self.ws)], "bn_stats", v=[raw], i=[code, B, 0, G], s=[bn.eps, bn.momentum],
                 g=[mean, invstd, rmean, rvar], p=[self.ws], n=[self.ws.numel()])
        if out is None:
            out = f.new_buf(raw.H, raw.W, raw.C).view()
        tw = self.q8_produced(f, out, 0)             # fp8 mode: the e4m3 twin of the output comes out of the same pass
        if tw is None:
            f.op(_hip.OP_BN_APPLY, [raw, res, self.T(mean), self.T(invstd)], [out], "bn_act", v=[raw, out, res], i=[code, B, act, G],
                 f=[mean, invstd, gamma, beta])
        else:
            f.op(_hip.OP_BN_APPLY, [raw, res, self.T(mean), self.T(invstd)], [out, tw[0]], "bn_act", v=[raw, out, res, tw[0]], i=[code, B, act, G],
                 f=[mean, invstd, gamma, beta], g=[None, None, tw[1], tw[2]])
            self.twin_only.append((f, lib().yp_plan_num_ops(f.handle) - 1, 1, out))       # (see _drop_unread_16bit_copies)

        def backward():
            b = self.bwd
            B = b.B                  # (the plan's samples: in pair mode the image pass's B for the YOLO-branch plan -- statistics group 0)
            gy = self.gread(out)
            fuse_res = res is not None
            gr = None
            if res is not None:
                gr, acc = self.gview(res)
                if not fuse_res:
                    b.op(_hip.OP_ADD_VIEWS, [gy, gr], [gr], "res_add", v=[gy, gr], i=[code, B, int(acc)])
            # (the shortcut gradient gr (+)= gy rides in the BatchNorm backward's row pass, which reads gy anyway: p1 / i5..i7 of YP_OP_BN_BWD)
            rs = dict(p1=gr.buf.t.data_ptr(), i567=[gr.cstride, gr.coff, int(acc)]) if fuse_res else dict(p1=None, i567=[0, 0, 0])
            rw = [gr] if fuse_res else []
            draw = b.new_buf(raw.H, raw.W, raw.C).view()
            gw_, gb_ = self.pgrad(bn.weight), self.pgrad(bn.bias)
            dg, db = (b.new_tensor((Cp,)), b.new_tensor((Cp,))) if padded else (gw_, gb_)
            # fp8 mode: the e5m2 twin of dx (the dgrad's operand) comes out of the same pass
            tw = self.q8_produced(b, draw, 1) if (sw("YP_FP8_DGRAD") != "0" and not image) else None
            if tw is None:
                b.op(_hip.OP_BN_BWD, [raw, gy] + rw, [draw, self.T(self.ws)] + rw, "bn_act_bwd", v=[raw, gy, draw], i=[code, B, act, 0, self.bG] + rs["i567"],
                     f=[mean, invstd, gamma, beta], g=[dg, db], p=[self.ws, rs["p1"]], n=[self.ws.numel()])
            else:
                b.op(_hip.OP_BN_BWD, [raw, gy] + rw, [draw, self.T(self.ws), tw[0]] + rw, "bn_act_bwd", v=[raw, gy, draw, tw[0]],
                     i=[code, B, act, 0, self.bG] + rs["i567"], f=[mean, invstd, gamma, beta], g=[dg, db, tw[1], tw[2]], p=[self.ws, rs["p1"]], n=[self.ws.numel()])
                self.twin_only.append((b, lib().yp_plan_num_ops(b.handle) - 1, 2, draw))
            if padded:
                self.collect.append(lambda dg=dg, db=db, gw_=gw_, gb_=gb_: (gw_.copy_(dg[:Cc]), gb_.copy_(db[:Cc])))
            self.conv_backward(srcs, conv.weight, None, draw, k, s, p, x8=x8)
        self.tape.append((self.branch, backward, self.tape_tag))
        return out

    def conv_plain(self, weight, bias, x, k, s, p, name):
        """Head convolution without BN/activation, fp32 output (ConvDet / ConvDesc)."""
        f = self.fwd
        out = f.conv([x], MasterWeight(weight, bias), bias, k, s, p, _hip.YP_ACT_NONE, out_f32=True)

        def backward():
            b = self.bwd
            g32 = self.gread(out)                       # fp32 NHWC gradient of the head output
            if self.code == _hip.YP_F32:
                draw = g32
            else:
                draw = b.new_buf(out.H, out.W, out.C).view()
                b.op(_hip.OP_CAST_F32, [g32], [draw], "cast", v=[g32, draw], i=[self.code, b.B])
            self.__dict__.setdefault("head_debug", {})[name] = dict(x=x, g32=g32, draw=draw, out=out)      # (probes: tools/probe/twin_only_dbg.py)
            for v_ in (x, draw):                        # the probe dictionary hands these views to torch-side readers: keep their 16-bit copies
                self.register_external_read(v_)
            self.conv_backward([x], weight, bias, draw, k, s, p)
        self.tape.append((self.branch, backward, self.tape_tag))
        return out

    def conv_backward(self, srcs, weight, bias, draw, k, s, p, x8=None):
        """Emit wgrad (+ bias grad) and dgrad of a convolution whose output gradient is `draw` [B,Ho,Wo,Cout_pad].
        x8 (fp8 mode): (e4m3 twin views of `srcs`, their scale slot) as the forward convolution read them."""
        b, code = self.bwd, self.code
        B = b.B
        Bpad = round_up(B, 8)
        Cout, Cout_pad = weight.shape[0], draw.C
        Ho, Wo = draw.H, draw.W
        K = Ho * Wo * Bpad
        assert K % 32 == 0, "wgrad needs Ho*Wo*round_up(B,8) to be a multiple of 32"
        gw = self.pgrad(weight)
        if bias is not None:
            gb_full = b.new_tensor((Cout_pad,))
            b.op(_hip.OP_COL_SUM, [draw], [self.T(gb_full), self.T(self.ws)], "bias_grad", v=[draw], i=[code, B, 0], g=[gb_full], p=[self.ws], n=[self.ws.numel()])
            if bias not in self.pgrads and self.vparts.get(id(bias)) is None:
                self.pgrads[bias] = gb_full[:Cout]         # the column sums ARE the bias gradient: no copy
                self.touched.add(bias)
            else:
                gbias = self.pgrad(bias)
                self.collect.append(lambda: gbias.copy_(gb_full[:Cout]))
        # ---- wgrad.  Stride-1 1x1 / 3x3 convolutions in a 16-bit dtype: yp_conv_wgrad reads the NHWC tensors directly
        # (LDS transpose reads; 3x3 also at stride 2).  Everything else (the 6x6 stem, fp32): wgrad as a convolution of pixel-major
        # copies -- the output gradient becomes the "filter" [Cout_pad (+1 zero row)][K].
        direct = code != _hip.YP_F32 and p == k // 2 and ((k == 1 and s == 1) or (k == 3 and s in (1, 2)))
        dyp = None
        dq = None
        # fp8 mode: the e5m2 copy of the output gradient (one per convolution: the operand of its sources' dgrads AND of its weight gradients)
        q8_dy = (self.fp8 and bias is None and draw.buf.t.dtype == torch.bfloat16 and self.q8_ok([draw]) and sw("YP_FP8_DGRAD") != "0")
        if q8_dy and all(src.C % 8 == 0 and not (src.geom is None and src.cstride == 4 and src.C == 4) for src in srcs):
            dq = self.q8_sources(b, [draw], 1)
        # 8-bit weight gradients (csrc/wgrad.hip::wgrad_body8): x = the e4m3 twin the forward convolution multiplied, dy = the e5m2 twin the dgrad
        # multiplies -- half the bytes of the 16-bit operands; the layers where both twins exist (all channel counts multiples of 64)
        q8_w = (x8 is not None and dq is not None and direct and self.group_wgrad and sw("YP_FP8_WGRAD") != "0"
                and all(v.C % 16 == 0 and v.coff % 16 == 0 and v.cstride % 16 == 0 for v in x8[0]))
        c0 = 0
        for j, src in enumerate(srcs):
            image = src.geom is None and src.cstride == 4 and src.C == 4
            Cj = src.C
            Hi, Wi = src.LH, src.LW
            ndw = Cj * k * k * Cout_pad
            assert self.dw_used + ndw <= self.dw_arena.numel()
            dwb = Buf(Cj, k, k, Cout_pad, torch.float32, self.device, storage=self.dw_arena[self.dw_used:self.dw_used + ndw])
            self.dw_used += round_up(ndw, 64)
            if direct and not image and self.group_wgrad:
                # nothing in the rest of the backward reads dW or overwrites x / dy: the weight gradients of a filter class are
                # collected and run as ONE grouped launch at the end of the pass (emit())
                # (a class = filter size, stride and the workgroup block size the entry runs with: one kernel instantiation per launch)
                if q8_w:
                    st8 = b.fp8
                    self.wgroups.setdefault((k, s, lib().yp_wgrad_block(src.c(), draw.c(), B, k), B, True), []).append(
                        (x8[0][j], dq[0][0], dwb, 2 * B * Ho * Wo * Cj * k * k * Cout, st8.scale_ptr(x8[1]), st8.scale_ptr(dq[1])))
                    self.n_q8w += 1
                else:
                    self.wgroups.setdefault((k, s, lib().yp_wgrad_block(src.c(), draw.c(), B, k), B, False), []).append(
                        (src, draw, dwb, 2 * B * Ho * Wo * Cj * k * k * Cout))
            elif image and code != _hip.YP_F32 and (k, s, p) == (6, 2, 2) and Cout_pad <= 80:
                # the stem: its own kernel over the packed image (no pixel-major copies of the two largest tensors of the pass)
                nsl = lib().yp_stem_wgrad_slabs(B, Hi, Wi)
                slabs = torch.empty((nsl, 144 * Cout_pad), dtype=torch.float32, device=self.device)
                self.keep.append(slabs)
                b.op(_hip.OP_STEM_WGRAD, [src, draw], [dwb.view(), self.T(slabs)], "wgrad_stem", v=[src, draw], i=[code, B], p=[slabs, dwb.flat])
                b.records[-1].kind, b.records[-1].flops = "conv", 2 * B * Ho * Wo * 3 * k * k * Cout
            elif direct and not image:
                b.op(_hip.OP_WGRAD, [src, draw], [dwb.view()], "wgrad", v=[src, draw], i=[code, B, k, s], p=[dwb.flat])
                b.records[-1].kind, b.records[-1].flops = "conv", 2 * B * Ho * Wo * Cj * k * k * Cout
                if self.lanes:       # runs beside the dgrad chain (second lane of the graph)
                    b.set_lane(_hip.LANE_SIDE)
            else:
                if dyp is None:
                    dyp = torch.zeros(((Cout_pad + 1) * K,), dtype=self.tdtype, device=self.device)
                    self.keep.append(dyp)
                    b.op(_hip.OP_TO_CHWB, [draw], [self.T(dyp)], "dy_chwb", v=[draw], i=[code, B, Cout_pad, Bpad], p=[dyp])
                xp = Buf(Cj, Hi, Wi, Bpad, self.tdtype, self.device)
                self.keep.append(xp.flat)
                b.op(_hip.OP_TO_CHWB, [src], [xp.view()], "x_chwb", v=[src], i=[code, B, Cj, Bpad], p=[xp.t])
                blocks = -(-(Cj * k * k) // 64) * -(-Cout_pad // 64)
                nk = K // (16 if code == _hip.YP_F32 else 32)
                ksplit = max(1, min(-(-1024 // blocks), max(1, nk // 8), 2048))
                if self.det_wgrad and ksplit > 1:
                    # deterministic split-K: one slab per k slice (plain stores), summed in slice order (the atomics of the default
                    # epilogue made the stem's weight gradient the one tensor that differed between two identical passes)
                    slabs = torch.empty((ksplit, ndw), dtype=torch.float32, device=self.device)
                    b.conv([xp.view()], None, None, 0, 1, p, _hip.YP_ACT_NONE, out=dwb.view(), out_f32=True,
                           extra=dict(raw_weight=(dyp, K, Cout_pad), cout=Cout_pad, kernel_hw=(Ho, Wo), dil=s, out_hw=(k, k), ksplit=ksplit,
                                      batch=Cj, weight_view=self.T(dyp), split_slabs=(slabs, ndw)))
                    b.op(_hip.OP_SUM_SLABS, [dwb.view()], [dwb.view()], "dw_fold", p=[slabs, dwb.flat], n=[ndw, ksplit])
                else:
                    b.conv([xp.view()], None, None, 0, 1, p, _hip.YP_ACT_NONE, out=dwb.view(), out_f32=True,
                           extra=dict(raw_weight=(dyp, K, Cout_pad), cout=Cout_pad, kernel_hw=(Ho, Wo), dil=s, out_hw=(k, k), ksplit=ksplit,
                                      atomic=1, batch=Cj, weight_view=self.T(dyp)))
            creal = weight.shape[1] - c0 if image else Cj
            # (dw -> OIHW gradient: all of a backward pass's transposes run as ONE launch at the end of the plan, see emit())
            self.unpack.append(dict(dw=dwb, grad=gw, rows=creal * k * k, cout=Cout, cout_pad=Cout_pad, out_stride=weight.shape[1] * k * k,
                                    out_off=c0 * k * k, split=1, pstride=0))
            # ---- dgrad (no gradient flows into the image)
            if not image:
                cs, ce_ = c0, c0 + Cj

                # dgrad = convolution with the flipped, channel-transposed filter [Cj, Cout_pad, k, k], packed on the device
                q8 = q8_dy and Cj % 8 == 0
                w_dgrad = MasterWeight(weight, mode=1, c0=cs, cj=Cj, cout_pad=Cout_pad, q8=q8)
                dsrc, dextra = draw, {}
                if q8:      # e5m2 copy of the output gradient (one per convolution, shared by its sources' dgrads) x e4m3 filter
                    if dq is None:
                        dq = self.q8_sources(b, [draw], 1)
                    dsrc, dextra = dq[0][0], dict(q8=dict(dtype=_hip.YP_FP8_BF8, slot=dq[1]))
                    self.n_q8 += 1
                base = View(src.buf, src.coff, src.C, 0, src.geom)
                if src.ups:
                    tmp = b.new_buf(Hi, Wi, Cj).view()
                    b.conv([dsrc], w_dgrad, None, k, 1, k - 1 - p, _hip.YP_ACT_NONE, out=tmp, extra=dict(zero_stuffed=(s == 2), **dextra))
                    gv, acc = self.gview(base)
                    b.op(_hip.OP_UPS2_BWD, [tmp, gv], [gv], "ups_bwd", v=[tmp, gv], i=[code, B, int(acc)])
                elif (s == 2 and k == 3 and p == 1 and Hi == 2 * Ho and Wi == 2 * Wo and sw("YP_DGRAD_PHASES") != "0"):
                    # stride 2: the input pixels of each parity class (py, px) receive only (1 + py) x (1 + px) of the nine taps -- four small
                    # stride-1 convolutions over dy that write their class of the gradient tensor, a quarter of the multiply-adds of one 3x3
                    # convolution over the zero-stuffed dy (YOLOPoint-l: 645 -> ~200 us for Conv2's dgrad)
                    gv, acc = self.gview(base)
                    for py in (0, 1):
                        for px in (0, 1):
                            gph = View(gv.buf, gv.coff, gv.C, 0, geom=(Ho, Wo, gv.cstride))
                            b.conv([dsrc], MasterWeight(weight, mode=("phase", py, px), c0=cs, cj=Cj, cout_pad=Cout_pad, q8=q8), None, 0, 1, 0,
                                   _hip.YP_ACT_NONE, out=gph, res=gph if acc else None, extra=dict(out_phase=(py, px), out_hw=(Ho, Wo), **dextra))
                else:
                    gv, acc = self.gview(base)
                    b.conv([dsrc], w_dgrad, None, k, 1, k - 1 - p, _hip.YP_ACT_NONE, out=gv, res=gv if acc else None,
                           extra=dict(zero_stuffed=(s == 2), **dextra))
            c0 += Cj

    def bottleneck(self, m, x, out=None):
        t = self.conv_bn_act(m.cv1, x)
        return self.conv_bn_act(m.cv2, t, out=out, res=x if m.add else None)

    def merged_siblings(self, a, b):
        """cv1 / cv2 of a C3 block read the same input through 1x1 convolutions: when their parameters and BatchNorm buffers lie back to
        back in memory (optim.FlatAdam's arena in training.grad_ready_groups order + training.link_siblings) they ARE one Conv with
        2c_ output channels -- returns that layer (a namespace with the attributes conv_bn_act reads), else None."""
        import types
        if type(a.act) is not type(b.act) or a.bn.eps != b.bn.eps or a.bn.momentum != b.bn.momentum:
            return None
        pairs = [(a.conv.weight, b.conv.weight), (a.bn.weight, b.bn.weight), (a.bn.bias, b.bn.bias), (a.bn.running_mean, b.bn.running_mean),
                 (a.bn.running_var, b.bn.running_var)]
        if a.conv.weight.shape != b.conv.weight.shape or a.conv.kernel_size != (1, 1) or a.conv.stride != (1, 1) or a.conv.bias is not None:
            return None
        if any(u.dtype != torch.float32 or not u.is_contiguous() or u.data_ptr() + 4 * u.numel() != v.data_ptr() or
               u.untyped_storage().data_ptr() != v.untyped_storage().data_ptr() for u, v in pairs):
            return None

        def both(u, v):
            t = torch.as_strided(u.detach(), (2 * u.shape[0],) + tuple(u.shape[1:]), u.stride())
            self.vparts[id(t)] = [u, v]
            self.keep.append(t)
            return t
        w, g_, b_ = (both(u, v) for u, v in pairs[:3])
        rm, rv = (torch.as_strided(u, (2 * u.shape[0],), (1,)) for u, v in pairs[3:])
        return types.SimpleNamespace(conv=types.SimpleNamespace(weight=w, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), out_channels=w.shape[0]),
                                     bn=types.SimpleNamespace(weight=g_, bias=b_, running_mean=rm, running_var=rv, eps=a.bn.eps, momentum=a.bn.momentum),
                                     act=a.act)

    def c3(self, m, x, out=None):
        c_ = m.cv1.conv.out_channels
        both = self.merged_siblings(m.cv1, m.cv2)
        if both is not None:
            # one convolution + one BatchNorm for cv1 and cv2 (forward: 1 conv + 2 BN launches instead of 2 + 4; backward: one BN backward,
            # one dgrad with K = 2c_ instead of two accumulating ones, one weight-gradient entry); cv3 reads [m(t), u] as two sources
            tu = self.conv_bn_act(both, x)
            t, u = tu.buf.view(tu.coff, c_), tu.buf.view(tu.coff + c_, c_)
            for blk in m.m:
                t = self.bottleneck(blk, t)
            return self.conv_bn_act(m.cv3, [t, u], out=out)
        x0 = x[0] if isinstance(x, (list, tuple)) else x
        cat = self.fwd.new_buf(x0.LH, x0.LW, 2 * c_)
        t = self.conv_bn_act(m.cv1, x)
        n = len(m.m)
        for i, blk in enumerate(m.m):
            t = self.bottleneck(blk, t, out=cat.view(0, c_) if i == n - 1 else None)
        self.conv_bn_act(m.cv2, x, out=cat.view(c_, c_))
        return self.conv_bn_act(m.cv3, cat.view(), out=out)

    def c2f(self, m, x, out=None):
        """C2f (reference models/common.py:151-171): cv1 -> channels [0, 2c) of the concat buffer; Bottleneckv8 i (two 3x3
        convs, optional shortcut) reads slice 1+i and writes slice 2+i; cv2 over the whole buffer."""
        c, n = m.c, len(m.m)
        x0 = x[0] if isinstance(x, (list, tuple)) else x
        cat = self.fwd.new_buf(x0.LH, x0.LW, (2 + n) * c)
        self.conv_bn_act(m.cv1, x, out=cat.view(0, 2 * c))
        for i, blk in enumerate(m.m):
            src = cat.view((1 + i) * c, c)
            t = self.conv_bn_act(blk.cv1, src)
            self.conv_bn_act(blk.cv2, t, out=cat.view((2 + i) * c, c), res=src if blk.add else None)
        return self.conv_bn_act(m.cv2, cat.view(), out=out)

    def maxpool2(self, x):
        f, code, B = self.fwd, self.code, self.B
        y = f.new_buf(x.LH // 2, x.LW // 2, x.C).view()
        f.op(_hip.OP_MAXPOOL2, [x], [y], "maxpool2", v=[x, y], i=[code, B])

        def backward():
            gy = self.gread(y)
            gx, acc = self.gview(x)
            self.bwd.op(_hip.OP_MAXPOOL2_BWD, [x, gy, gx], [gx], "maxpool2_bwd", v=[x, gy, gx], i=[code, self.bwd.B, int(acc)])
        self.tape.append((self.branch, backward, self.tape_tag))
        return y

    def sppf(self, m, x):
        c_ = m.cv1.conv.out_channels
        f, code, B = self.fwd, self.code, self.B
        cat = f.new_buf(x.LH, x.LW, 4 * c_)
        s0, s1, s2, s3 = (cat.view(i * c_, c_) for i in range(4))
        self.conv_bn_act(m.cv1, x, out=s0)
        f.op(_hip.OP_SPPF_POOL, [s0], [s1, s2, s3], "pool", v=[s0, s1, s2, s3], i=[code, B])

        def backward():
            b = self.bwd
            B = b.B
            g0, g1, g2, g3 = (self.gread(v) for v in (s0, s1, s2, s3))
            b.op(_hip.OP_MAXPOOL5_BWD, [s2, g3, g2], [g2, self.T(self.ws)], "pool_bwd3", v=[s2, g3, g2], i=[code, B, 1], p=[self.ws], n=[self.ws.numel()])
            b.op(_hip.OP_MAXPOOL5_BWD, [s1, g2, g1], [g1, self.T(self.ws)], "pool_bwd2", v=[s1, g2, g1], i=[code, B, 1], p=[self.ws], n=[self.ws.numel()])
            b.op(_hip.OP_MAXPOOL5_BWD, [s0, g1, g0], [g0, self.T(self.ws)], "pool_bwd1", v=[s0, g1, g0], i=[code, B, 1], p=[self.ws], n=[self.ws.numel()])
        self.tape.append((self.branch, backward, self.tape_tag))
        return self.conv_bn_act(m.cv2, cat.view())

    # ------------------------------------------------------------------ the graph
    def _build(self):
        net, f, b, B, code = self.net, self.fwd, self.bwd, self.B, self.code
        Hc, Wc = self.H // 8, self.W // 8
        v52 = type(net).__name__ == "YOLOPointv52"
        blk = self.c2f if v52 else self.c3
        self.img = f.new_buf(self.H, self.W, 4)
        x = self.conv_bn_act(net.Conv1, self.img.view())
        x = self.conv_bn_act(net.Conv2, x)
        xa = blk(net.Bottleneck1, x)
        x8 = self.conv_bn_act(net.Conv3, xa)
        # keypoint head: C3 + plain 1x1 conv (fp32) | v52: a 65-channel C2f whose BN + SiLU output IS semi
        # Forward lanes (YP_TRAIN_FWD_LANES=0 turns them off): the two heads on the forward plan's side lane, beside the YOLO encoder / PAN /
        # Detect chain (whose P4 / P5 layers leave most CUs idle); the plan then replays eagerly on two streams (see PlanBuilder.side)
        fwd_lanes = sw("YP_TRAIN_FWD_LANES") != "0" and not self.lanes
        # Each head is emitted (= forked) right behind the last tensor it reads: Conv3 (x8) for the keypoint head, Bottleneck2 (xb) for the
        # descriptor head.  Later forks measured slower (both heads behind Bottleneck2 / 3 / 4: 7.60 / 7.84 / 8.01 ms per step).
        hd = {}

        def head_view(gv, C_):
            # The head gradients arrive from autograd as [B,C,H,W]-shaped tensors (usually already channels-innermost in memory: the heads
            # are handed out as permuted views).  backward() copies them straight into the NHWC gradient buffers through permuted views
            # (Tensor.copy_ converts layout and dtype in one pass) -- no NCHW staging buffer and no pack launch (2 x 95 us per step).
            return gv.buf.t[..., gv.coff:gv.coff + C_].permute(0, 3, 1, 2)

        def semi_seed():
            gsemi_v, _ = self.gview(hd["semi"])
            self.seed_semi = head_view(gsemi_v, 65)

        def desc_seed():
            hd["desc_seed"]()

        def emit_kp_head():
            self.tape_tag = "kph"
            with self.side_lane(f, fwd_lanes):
                if v52:
                    hd["semi"] = self.c2f(net.BottleneckDet, x8)
                else:
                    t = self.c3(net.BottleneckDet, x8)
                    hd["semi"] = self.conv_plain(net.ConvDet.weight, None, t, 1, 1, 0, "ConvDet")
            self.tape_tag = None

        def emit_desc_head():
            with self.side_lane(f, fwd_lanes):
                if v52:
                    # MaxPool(xa) ++ up(ConvDescB(xb)) -> C2f; the L2 normalisation is applied (and differentiated) by the caller in PyTorch
                    dA = self.maxpool2(xa)
                    dB = self.conv_bn_act(net.ConvDescB, xb)
                    craw = self.c2f(net.BottleneckDesc, [dA, dB.up()])
                    c3ch = net._desc_channels
                    dnorm = craw

                    def seed():
                        gcraw, _ = self.gview(craw)
                        self.seed_desc = head_view(gcraw, c3ch)
                else:
                    dA = self.conv_bn_act(net.ConvDescA, xa)
                    dB = self.conv_bn_act(net.ConvDescB, xb)
                    d = self.c3(net.BottleneckDesc, [dA, dB.up()])
                    craw = self.conv_plain(net.ConvDesc.weight, None, d, 3, 1, 1, "ConvDesc")
                    c3ch = net.ConvDesc.out_channels
                    dnorm = f.new_buf(Hc, Wc, craw.C, f32=True).view()
                    f.op(_hip.OP_L2NORM, [craw], [dnorm], "l2norm", v=[craw, dnorm], i=[0, B, c3ch])
                    gd = Buf(B, Hc, Wc, craw.C, torch.float32, self.device)
                    self.keep.append(gd.flat)

                    def seed():
                        b = self.bwd
                        gcraw, _ = self.gview(craw)
                        self.seed_desc = head_view(gd.view(), c3ch)
                        self.seed_desc_buf = gd
                        b.op(_hip.OP_L2NORM_BWD, [craw, gd.view()], [gcraw], "l2norm_bwd", v=[craw, gd.view(), gcraw], i=[0, b.B, c3ch])
            hd["desc_seed"], hd["dnorm"], hd["c3ch"] = seed, dnorm, c3ch
        emit_kp_head()
        xb = blk(net.Bottleneck2, x8)
        emit_desc_head()
        # YOLO encoder + PAN: nothing below feeds semi / desc
        self.branch = "yolo"
        self.kp_ptrs = set(t_.data_ptr() for t_ in f.keep)        # the trunk / keypoint-branch buffers (gview: pair mode)
        x = self.conv_bn_act(net.Conv4, xb)
        xc = blk(net.Bottleneck3, x)
        x = self.conv_bn_act(net.Conv5, xc)
        x = blk(net.Bottleneck4, x)
        x = self.sppf(net.SPPooling, x)
        if v52:
            xd = x
            xe = blk(net.Bottleneck5, [xd.up(), xc])
            xf = blk(net.Bottleneck6, [xe.up(), xb])
        else:
            xd = self.conv_bn_act(net.Conv6, x)
            x = self.c3(net.Bottleneck5, [xd.up(), xc])
            xe = self.conv_bn_act(net.Conv7, x)
            xf = self.c3(net.Bottleneck6, [xe.up(), xb])
        # Detect (train mode: permuted raw logits only).  YP_TRAIN_DET_LANES=1 puts the P3 / P4 levels on the side lane as the inference plan
        # does -- measured slower here (8.12 vs 7.37 ms per step: they queue behind the heads on the one side stream and the object loss,
        # the head of the backward's critical path, waits for them), so they stay on the main lane.
        det = net.Detect
        self.xs, self.g_xs = [], []
        det_seeds = []

        def detect_level(i, v):
            ny, nx = v.LH, v.LW
            xo = f.new_tensor((B, det.na, ny, nx, det.no))
            stride = float(det.stride[i])
            mi = det.m[i]
            f.conv(v, MasterWeight(mi.weight, mi.bias), mi.bias, 1, 1, 0, _hip.YP_ACT_NONE, out_f32=True,      # (packed on the device with all other filters)
                   detect=dict(na=det.na, no=det.no, stride=stride, anchors_px=[0.0] * (2 * det.na), x_out=xo, z_out=None, rows_total=0,
                               row_offset=0))
            gx = torch.zeros_like(xo)
            self.xs.append(xo)
            self.g_xs.append(gx)

            def det_backward(v=v, mi=mi, gx=gx, ny=ny, nx=nx):
                b = self.bwd
                draw = b.new_buf(ny, nx, round_up(det.na * det.no, 8)).view()
                b.op(_hip.OP_DETECT_BWD_PACK, [self.T(gx)], [draw], "seed_det", f=[gx, self.head_scale], v=[draw], i=[code, b.B, det.na, det.no])
                self.conv_backward([v], mi.weight, mi.bias, draw, 1, 1, 0)
            det_seeds.append(det_backward)
        # Detect levels 0 / 1 (a 1x1 convolution + its epilogue each) on the side lane beside Conv8 / Conv9 of the PAN.  Round 5, after the convolution
        # epilogues stopped waiting for their own stores: -s step 7.19 -> 6.83 ms at batch 8 (same box, two pairs; 7.02 -> 6.79 with the previous
        # library), -l and batch 64 unchanged (33.4 / 33.4, 29.6 / 29.7, 41.8 / 42.1 ms).  YP_TRAIN_DET_LANES=0: off.  Same kernels, same bits.
        det_lanes = fwd_lanes and sw("YP_TRAIN_DET_LANES") == "1"
        with self.side_lane(f, det_lanes):
            detect_level(0, xf)
        x = self.conv_bn_act(net.Conv8, xf)
        xg = blk(net.Bottleneck7, [x, xe])
        with self.side_lane(f, det_lanes):
            detect_level(1, xg)
        x = self.conv_bn_act(net.Conv9, xg)
        p5 = blk(net.Bottleneck8, [x, xd])
        detect_level(2, p5)
        self.semi_v, self.desc_v, self.desc_channels = hd["semi"], hd["dnorm"], hd["c3ch"]
        # YP_TRAIN_PARALLEL=1: replay the launch lists as the DAG of their data dependencies (multi-kernel ops keep their inner chain) instead
        # of linear chains.  Measured SLOWER for the training step (14.4 vs 13.1 ms: the concurrent BatchNorm / weight-gradient / dgrad
        # kernels fight over CUs and L2, as the two-lane schedule and the sub-batch streams of the forward did), so it is off by default;
        # bit-identical results either way (tests/test_gpu_training.py).
        self.par = not self.lanes and sw("YP_TRAIN_PARALLEL") == "1"
        self.fwd_plan = f.finish(parallel=self.par)

        # ---- backward plans: clear the weight-gradient arena, seed the head gradients, then the tape in reverse.
        # Two variants: the full one, and one that only back-propagates the semi / desc sub-graph -- the reference's second
        # forward of a step (warped image) has no object loss, so autograd never visits its Detect / PAN / YOLO-encoder
        # layers (4 F_fwd + 2 F_kp per sample, not 6 F_fwd).
        def emit(branches, B, groups, fresh=True):
            """One backward plan over the tape entries of `branches` ('kp': trunk + keypoint / descriptor heads, 'yolo': YOLO encoder + PAN +
            Detect), B samples, `groups` statistics groups.  fresh=False: the activation gradients an earlier plan of the same pass wrote
            stay valid (pair mode: the trunk plan accumulates onto what the YOLO-branch plan left in the backbone output's gradient)."""
            self.bwd = bb = PlanBuilder(B, code, self.device)
            self.builders.append(bb)
            bb.fp8 = self.fwd.fp8
            self.bG = groups
            bb.pack_target = self.fwd.pack_target
            if fresh:
                for key in self.gwritten:
                    self.gwritten[key] = []
            self.touched, self.collect, self.unpack, self.wgroups = set(), [], [], {}
            if self.G > 1:
                self.wpart = None         # (pair mode: one slab arena per plan -- the two plans differ in batch)
            self.dw_used = 0              # (the plans of a graph run one after another and unpack their accumulators at their end)
            bb.op(_hip.OP_MEMSET0, [], [self.T(self.dw_arena)], "zero_dw", p=[self.dw_arena], n=[self.dw_arena.numel() * 4])
            if "kp" in branches:
                semi_seed()
                desc_seed()
            if "yolo" in branches:
                for fn in det_seeds:      # Detect backward runs before the PAN blocks' backward (it writes their output gradients)
                    fn()
            for branch, fn, tag in reversed(self.tape):
                if branch in branches:
                    fn()
            for (gk, gs, gblk, gB, g8), ents in sorted(self.wgroups.items()):
                n = len(ents)
                gcode = _hip.YP_FP8 if g8 else code       # (8-bit entries: x / dy are the 1-byte twins, e[4] / e[5] their scale pointers)
                xs, dys = (_hip.YpView * n)(*[e[0].c() for e in ents]), (_hip.YpView * n)(*[e[1].c() for e in ents])
                dws = (C.c_void_p * n)(*[e[2].flat.data_ptr() for e in ents])
                host = (C.c_char * (n * lib().yp_wgrad_group_entry_bytes()))()
                blocks, fold = C.c_int(0), C.c_int(0)
                parts = None
                if self.det_wgrad:
                    # deterministic reduction: every pixel slice of an entry writes its own slab, a second launch sums them in order.
                    # The slab arena is shared by the filter classes of all backward plans of this graph (they run one after another).
                    sizes = [round_up(lib().yp_wgrad_partial_elems(e[0].c(), e[1].c(), gcode, gB, gk, gs, gblk), 64) for e in ents]
                    if self.wpart is None:      # sized by the largest filter class of the FULL backward (emitted first; the keypoint-only plan is a subset)
                        self.wpart = torch.empty(max(sum(round_up(lib().yp_wgrad_partial_elems(e[0].c(), e[1].c(), _hip.YP_FP8 if q_ else code, B_, k_, s_, b_), 64)
                                                         for e in es) for (k_, s_, b_, B_, q_), es in self.wgroups.items()), dtype=torch.float32, device=self.device)
                        self.keep.append(self.wpart)
                    assert sum(sizes) <= self.wpart.numel()
                    offs = [sum(sizes[:i]) for i in range(n)]
                    parts = (C.c_void_p * n)(*[self.wpart.data_ptr() + 4 * o for o in offs])
                if g8:
                    sxs, sdys = (C.c_void_p * n)(*[e[4] for e in ents]), (C.c_void_p * n)(*[e[5] for e in ents])
                    check(lib().yp_wgrad_group_pack_q8(xs, dys, dws, parts, sxs, sdys, n, gB, gk, gs, gblk, host, C.byref(blocks), C.byref(fold) if parts is not None else None))
                elif self.det_wgrad:
                    check(lib().yp_wgrad_group_pack_det(xs, dys, dws, parts, n, code, gB, gk, gs, gblk, host, C.byref(blocks), C.byref(fold)))
                else:
                    check(lib().yp_wgrad_group_pack(xs, dys, dws, n, code, gB, gk, gs, gblk, host, C.byref(blocks)))
                wtab = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
                self.keep.append(wtab)
                bb.op(_hip.OP_WGRAD_GROUP, [v for e in ents for v in e[:2]], [e[2].view() for e in ents] + ([self.T(self.wpart)] if self.det_wgrad else []),
                      f"wgrad_k{gk}s{gs}" + ("b128" if gblk == 128 else "") + ("q8" if g8 else ""), p=[wtab],
                      i=[gcode, n, blocks.value, gk, gs, fold.value, gblk])
                bb.records[-1].kind, bb.records[-1].flops = "conv", sum(e[3] for e in ents)
            rows, tile0 = [], 0
            for u in self.unpack:
                rows.append([u["dw"].flat.data_ptr(), u["grad"].data_ptr(), u["rows"], u["cout"], u["cout_pad"], u["out_stride"], u["out_off"], tile0,
                             u["split"], u["pstride"]])
                tile0 += -(-u["rows"] // 32) * -(-u["cout"] // 32)
            table = torch.tensor(rows, dtype=torch.int64).to(self.device)
            self.keep.append(table)
            bb.op(_hip.OP_WGRAD_UNPACK_BATCH, [u["dw"].view() for u in self.unpack], [self.T(u["grad"]) for u in self.unpack], "dw_unpack",
                  p=[table], i=[0, len(rows), tile0])
            bb.set_lane(_hip.LANE_JOIN)
            plan = bb.finish(parallel=self.par)
            return plan, self.touched, self.collect
        if self.G == 1:
            self.bwd_plan, self.bwd_params, self.bwd_collect = emit(("kp", "yolo"), B, 1)
            self.bwd_kp_plan, self.bwd_kp_params, self.bwd_kp_collect = emit(("kp",), B, 1)
        else:
            # pair mode: the YOLO-branch layers over the image pass's samples (statistics group 0 = the first Bs samples of every buffer),
            # then the trunk + keypoint / descriptor heads over both passes
            # (Round 4 built a variant with the keypoint head's backward on the YOLO-branch plan's side lane, YP_TRAIN_BWD_LANES: measured slower,
            # 7.60 vs 7.40 ms per step, and round 6's switch harness found that its gradients differ from the default's on 24 tensors -- removed.)
            self.bwd_plan, self.bwd_params, self.bwd_collect = emit(("yolo",), self.Bs, 1)
            self.bwd_kp_plan, self.bwd_kp_params, self.bwd_kp_collect = emit(("kp",), B, self.G, fresh=False)
        self._drop_unread_16bit_copies(net)
        mode = sw("YP_TRAIN_GRAPH")         # replay the launch lists as hipGraphs (284 / 455 / 230 launches)
        if mode in ("1", "fwd") and not (self.fwd_plan.has_lanes and sw("YP_LANES_EAGER") != "0"):
            self.fwd_plan.instantiate_graph()
        if mode in ("1", "bwd"):
            if not (self.bwd_plan.has_lanes and sw("YP_LANES_EAGER") != "0"):
                self.bwd_plan.instantiate_graph()
            self.bwd_kp_plan.instantiate_graph()
        self.params = [p_ for p_ in net.parameters()]

    def _drop_unread_16bit_copies(self, net):
        """fp8 mode, after ALL plans of the graph have been emitted: a BatchNorm pass that writes a 1-byte twin beside its 16-bit result (the
        activation in the forward, dy of the convolution in the backward) stops writing the 16-bit copy when nothing reads it -- every
        consumer is an 8-bit convolution / dgrad / weight gradient (csrc/wgrad.hip::wgrad_body8), which read the twin.  Readers are taken from
        the access lists of every plan builder of the graph (the lists behind the hipGraph dependency edges); residual inputs, pools, Detect /
        head convolutions with a bias, 16-bit layers (channel counts that are not multiples of 64) keep their 16-bit source.  An unread copy is
        filled with NaN once: a reader this analysis missed cannot go unnoticed (non-finite loss / gradients in the fp8 tests).
        3 -> 1 bytes written per element in those passes.  YP_FP8_TWIN_ONLY=0: keep every copy.
        The opt-in is structural: a model class declares `plans_cover_all_reads = True` when every reader of its training buffers is a plan op
        (YOLOPoint; the default of HipModule is False, so a subclass / another model that differentiates anything in PyTorch -- YOLOPointv52's
        descriptor normalisation -- keeps every copy unless it says otherwise), and whatever reads a view from outside the plans registers it
        (register_external_read: the probe dictionary head_debug does).  Dropped views are recorded (dropped_views / assert_readable), and the
        first fp8 optimizer step of engine.TrainStep checks the gradients for non-finite values."""
        if not self.fp8 or not self.twin_only or sw("YP_FP8_TWIN_ONLY") == "0" or not type(net).__dict__.get("plans_cover_all_reads", False):       # (the class's OWN declaration: not inherited)
            return
        reads = [r for pb in [self.fwd] + self.builders for rd, _ in pb.accesses for r in rd] + [PlanBuilder._rng(v_) for v_ in self.external_reads]

        def overlap(a, b):
            if a[0] != b[0]:
                return False
            if a[1] == b[1]:
                return a[2] < b[3] and b[2] < a[3]
            return a[4] < b[5] and b[4] < a[5]
        for pb, op, slot, view in self.twin_only:
            rng = PlanBuilder._rng(view)
            if any(overlap(rng, r) for r in reads):
                continue
            check(lib().yp_plan_patch_op_view(pb.handle, op, slot, NULL_VIEW))
            view.buf.t[..., view.coff:view.coff + view.C] = float("nan")
            self.n_twin_only += 1
            self.dropped_views.append(view)

    def register_external_read(self, view):
        """Declare that code outside the plans (torch-side differentiation, a probe, an exporter) reads `view`: its 16-bit copy is kept in fp8
        mode.  Must be called while the graph is emitted (before _drop_unread_16bit_copies)."""
        if self.dropped_views:
            raise _hip.YpError("register_external_read after the unread 16-bit copies were dropped: build the graph with YP_FP8_TWIN_ONLY=0")
        self.external_reads.append(view)

    def assert_readable(self, view):
        """For accessors that hand a graph buffer to torch-side code (probes, debuggers): raises when the view's 16-bit copy is one the fp8
        graph no longer writes (its memory holds NaN)."""
        rng = PlanBuilder._rng(view)
        for d in self.dropped_views:
            r = PlanBuilder._rng(d)
            if r[0] == rng[0] and ((r[1] == rng[1] and r[2] < rng[3] and rng[2] < r[3]) or (r[1] != rng[1] and r[4] < rng[5] and rng[4] < r[5])):
                raise _hip.YpError("this activation's 16-bit copy is not written in fp8 twin-only mode (only its 1-byte twin is): "
                                   "set YP_FP8_TWIN_ONLY=0 for tools that read activations")
        return view

    # ------------------------------------------------------------------ run
    def forward(self, x, x_w=None, export=True):
        """x: [B,3,H,W] fp32 (pair mode: x = the image batch, x_w = the warped image batch; the heads come back for all 2B samples, image
        pass first; the Detect levels for the image pass only).  export=False: nothing is returned -- the caller reads the heads where
        the plan left them (semi_v / desc_v buffers, xs) and writes their gradients where the backward plans read them (seed_semi,
        seed_desc_buf, g_xs), then calls backward_pair with SEEDED."""
        assert (x_w is not None) == (self.G > 1)
        # packed weights (forward + dgrad) are re-derived only when an optimizer step (or a load) changed the masters
        # (the optimizer-step count is part of the key: fused optimizers do not bump Tensor._version)
        ver = (weights_generation(),) + tuple(p_._version for p_ in self.params)      # (a tuple: a sum of counters can collide)
        if ver != getattr(self, "_packed_version", None):      # host-packed filters of this graph (stem, Detect)
            self.fwd_plan.refresh()
            self.bwd_plan.refresh()
            self.bwd_kp_plan.refresh()
            self._packed_version = ver
        nops = lib().yp_plan_num_ops(self.pack["pb"].handle)
        ents8 = getattr(self.pack["pb"], "pack8_entries", [])
        if (ver, nops, len(ents8)) != self.pack["version"]:     # device-packed filters shared by all graphs: one batched launch
            if "fp8" in self.pack:
                # once per optimizer step: last step's recorded maxima become this step's quantisation scales (delayed scaling), then
                # the e4m3 filter copies are re-derived with their new scales
                self.pack["fp8"].update(float(sw("YP_FP8_MARGIN")))
            ents = getattr(self.pack["pb"], "pack_entries", [])
            if len(ents) == nops:
                if self.pack.get("table_n") != nops:
                    rows, blk0 = [], 0
                    for e in ents:
                        rows.append(e + [blk0])
                        blk0 += -(-((e[13] + 1) * e[12]) // 1024)
                    self.pack["table"], self.pack["table_n"], self.pack["blocks"] = torch.tensor(rows, dtype=torch.int64).to(self.device), nops, blk0
                check(lib().yp_pack_weight_batch(self.pack["table"].data_ptr(), nops, self.pack["blocks"], self.code, _hip.stream_ptr()))
            elif nops:
                check(lib().yp_plan_run(self.pack["pb"].handle, _hip.stream_ptr()))
            if ents8:
                if self.pack.get("table8_n") != len(ents8):
                    rows, blk0 = [], 0
                    for e in ents8:
                        rows.append(e + [blk0])
                        blk0 += -(-((e[13] + 1) * e[12]) // 1024)
                    self.pack["table8"], self.pack["table8_n"], self.pack["blocks8"] = torch.tensor(rows, dtype=torch.int64).to(self.device), len(ents8), blk0
                check(lib().yp_pack_weight_fp8_batch(self.pack["table8"].data_ptr(), len(ents8), self.pack["blocks8"], _hip.stream_ptr()))
            self.pack["version"] = (ver, nops, len(ents8))
        for fn in self.pre_forward:
            fn()
        if self.G == 1:
            pack_input(x, self.img.view(), self.code)
        else:
            for j, xx in enumerate((x, x_w)):
                vc = self.img.view().c()
                vc.ptr += j * self.Bs * self.H * self.W * 4 * self.img.t.element_size()
                assert xx.is_cuda and xx.dtype == torch.float32 and xx.is_contiguous() and xx.shape[0] == self.Bs
                check(lib().yp_pack_input(xx.data_ptr(), self.Bs, xx.shape[1], self.H, self.W, vc, self.code, _hip.stream_ptr()))
        self.fwd_plan.run()
        for fn in self.post_forward:
            fn()
        if self._nbt is None:
            self._nbt = [m.num_batches_tracked for m in self.net.modules()
                         if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None]
        if self._nbt:                                  # one launch for all BatchNorm counters (through a table of their addresses)
            if getattr(self, "_nbt_table", None) is None:
                assert all(t.dtype == torch.int64 and t.is_cuda for t in self._nbt)
                self._nbt_table = torch.tensor([t.data_ptr() for t in self._nbt], dtype=torch.int64).to(self.device)
            check(lib().yp_counters_add(self._nbt_table.data_ptr(), len(self._nbt), self.G, _hip.stream_ptr()))
        if not export:
            return None
        c3ch = self.desc_channels
        semi = self.semi_v.buf.t[..., :65].permute(0, 3, 1, 2).float()      # (a copy: fp32 heads are cloned, 16-bit ones converted)
        desc = self.desc_v.buf.t[..., :c3ch].permute(0, 3, 1, 2).float()
        if semi.data_ptr() == self.semi_v.buf.t.data_ptr():
            semi = semi.clone()
        if desc.data_ptr() == self.desc_v.buf.t.data_ptr():
            desc = desc.clone()
        return semi, desc, [t[:self.Bs].clone() for t in self.xs]

    def set_head_scale(self, val):
        """The factor the backward plan applies to the Detect-level seeds (a device scalar: the plans are replayed as hipGraphs)."""
        if self._head_scale_val != val:
            self.head_scale[0:1].fill_(val)
            self._head_scale_val = val

    def backward_pair(self, g_semi, g_desc, g_xs, between=None):
        """Pair mode: head gradients (semi / desc over the 2B samples, the Detect levels over the image pass's B; None = zero) ->
        parameter gradients.  Runs the YOLO-branch plan, calls between(parameter -> gradient of the parameters it reached) -- they are
        final: a data-parallel step starts their all-reduce here --, then the trunk plan; returns its dict."""
        assert self.G > 1
        if not all(src is SEEDED for src in g_xs):
            self.set_head_scale(1.0)                 # (gradients that arrive from autograd carry their upstream factor already)
        for dst, src in zip(self.g_xs, g_xs):
            if src is SEEDED:
                continue
            if src is None:
                dst[:self.Bs].zero_()
            else:
                dst[:self.Bs].copy_(src)
        def seed(dst, src):
            if src is SEEDED:
                return
            if src is None:
                dst.zero_()
            else:
                dst.copy_(src)
        self.bwd_plan.run()
        for fn in self.bwd_collect:
            fn()
        first = {p_: self.pgrads[p_] for p_ in self.params if p_ in self.bwd_params}
        if between is not None:
            between(first)
        seed(self.seed_semi, g_semi)
        seed(self.seed_desc, g_desc)
        self.bwd_kp_plan.run()
        for fn in self.bwd_kp_collect:
            fn()
        second = {p_: self.pgrads[p_] for p_ in self.params if p_ in self.bwd_kp_params}
        if between is None:
            second.update(first)
        return second

    def backward(self, g_semi, g_desc, g_xs):
        """Head gradients (None = that head took no part in the loss) -> parameter gradients (None = not reached)."""
        if self.G > 1:
            got = self.backward_pair(g_semi, g_desc, g_xs)
            return [got.get(p_) for p_ in self.params]
        kp_only = all(g is None for g in g_xs)
        self.set_head_scale(1.0)
        for dst, src in [(self.seed_semi, g_semi), (self.seed_desc, g_desc)] + ([] if kp_only else list(zip(self.g_xs, g_xs))):
            if src is None:
                dst.zero_()
            else:
                dst.copy_(src)
        plan, touched, collect = (self.bwd_kp_plan, self.bwd_kp_params, self.bwd_kp_collect) if kp_only else (self.bwd_plan, self.bwd_params, self.bwd_collect)
        plan.run()
        for fn in collect:
            fn()
        return [self.pgrads[p_] if p_ in touched else None for p_ in self.params]


SEEDED = object()       # backward_pair: "this head's gradient is already in the plan's seed buffer"


def _release(g):
    g.busy = False


class _YOLOPointTrainFn(torch.autograd.Function):
    @staticmethod
    @_hip.guarded
    def forward(ctx, net, x, *params):
        g = net._train_graph(x, fp8=bool(getattr(net, "fp8_train", False)))
        ctx.graph = g
        ctx.set_materialize_grads(False)       # a head that took no part in the loss arrives as None
        semi, desc, xs = g.forward(x)
        if torch.is_grad_enabled() or any(p.requires_grad for p in params):
            g.busy = True
            # a forward whose graph is dropped without a backward (e.g. a train-mode evaluation pass) must give its plans back
            weakref.finalize(ctx, _release, g)
        return (semi, desc, *xs)

    @staticmethod
    def backward(ctx, g_semi, g_desc, *g_xs):
        g = ctx.graph
        with torch.cuda.device(g.device):
            return _YOLOPointTrainFn._backward(ctx, g, g_semi, g_desc, g_xs)

    @staticmethod
    def _backward(ctx, g, g_semi, g_desc, g_xs):
        if getattr(g.net, "autograd_param_grads", False):
            # Opt-in (net.autograd_param_grads = True): the parameter gradients go back THROUGH autograd -- one tensor per parameter, as
            # loss.backward() of the reference produces them (train.py:245) -- so that AccumulateGrad hooks fire: the model can be wrapped in
            # torch.nn.parallel.DistributedDataParallel / accelerate.prepare (train.py:44-46,174), torch.autograd.grad works and
            # no_sync() behaves.  Copies of the plan's gradient buffers (the plans own those and overwrite them on the next pass).
            grads = g.backward(g_semi, g_desc, list(g_xs))
            g.busy = False
            live = [(i, gr) for i, (p_, gr) in enumerate(zip(g.params, grads)) if gr is not None and p_.requires_grad]
            fresh = [torch.empty_like(g.params[i]) for i, _ in live]
            if fresh:
                torch._foreach_copy_(fresh, [gr for _, gr in live])
            out = [None] * len(g.params)
            for (i, _), f_ in zip(live, fresh):
                out[i] = f_
            return (None, None, *out)
        run_native_backward(g, g_semi, g_desc, list(g_xs))
        return (None, None, *([None] * len(g.params)))


class _YOLOPointPairFn(torch.autograd.Function):
    """Both forwards of a training step in one launch list (TrainGraph pair mode)."""
    @staticmethod
    @_hip.guarded
    def forward(ctx, net, x, x_w, *params):
        g = net._train_graph(x, pair=True, fp8=bool(getattr(net, "fp8_train", False)))
        ctx.graph = g
        ctx.set_materialize_grads(False)
        semi, desc, xs = g.forward(x, x_w)
        if torch.is_grad_enabled() or any(p.requires_grad for p in params):
            g.busy = True
            weakref.finalize(ctx, _release, g)
        return (semi, desc, *xs)

    @staticmethod
    def backward(ctx, g_semi, g_desc, *g_xs):
        g = ctx.graph
        with torch.cuda.device(g.device):
            run_native_backward_pair(g, g_semi, g_desc, list(g_xs))
        return (None, None, None, *([None] * len(g.params)))


def run_native_backward(g, g_semi, g_desc, g_xs):
    """One native backward pass of TrainGraph `g` (head gradients; None = that head took no part in the loss), its parameter
    gradients accumulated into p.grad.  Parameter gradients are delivered with multi-tensor ops instead of ~215 per-parameter
    autograd returns (each of which AccumulateGrad would clone): a fresh copy for parameters without a gradient yet, an in-place
    add otherwise (p.grad may be a view of a gradient all-reduce bucket: dp.GradAllReducer.bind_grads).  Returns the parameters
    that received a contribution."""
    grads = g.backward(g_semi, g_desc, list(g_xs))
    g.busy = False
    return _deliver(g.params, grads)


def run_native_backward_pair(g, g_semi, g_desc, g_xs, notify=None, join=None):
    """The backward of a pair-mode TrainGraph (both passes of a training step), parameter gradients accumulated into p.grad as
    run_native_backward does.  notify(parameters) is called twice: after the YOLO-branch plan (the detector-group parameters are final
    while the trunk plan is still to run -- dp.GradAllReducer launches their buckets there) and after the trunk plan.  join() is called
    in front of the trunk plan: a caller that fills the semi / desc seeds on another stream makes the current stream wait there."""
    def between(first):
        ps = list(first)
        done = _deliver(ps, [first[p_] for p_ in ps])
        if notify is not None:
            notify(done)
        if join is not None:
            join()
    rest = g.backward_pair(g_semi, g_desc, list(g_xs), between=between)
    g.busy = False
    ps = list(rest)
    done = _deliver(ps, [rest[p_] for p_ in ps])
    if notify is not None:
        notify(done)
    return done


def _deliver(params, grads):
    new_p, new_g, acc_p, acc_g, touched = [], [], [], [], []
    for p_, gr in zip(params, grads):
        if gr is None or not p_.requires_grad:
            continue
        touched.append(p_)
        if p_.grad is None:
            new_p.append(p_); new_g.append(gr)
        else:
            acc_p.append(p_.grad); acc_g.append(gr)
    if new_p:
        fresh = [torch.empty_like(p_, dtype=p_.dtype) for p_ in new_p]
        torch._foreach_copy_(fresh, new_g)
        for p_, f_ in zip(new_p, fresh):
            p_.grad = f_
    if acc_p:
        _multi_add(acc_p, acc_g)
    return touched


_ADD_TABLES = {}


def _multi_add(dsts, srcs):
    """dst += src for a list of fp32 tensor pairs in ONE native launch (yp_multi_add) through a device table that is built once per
    distinct list (the plans' gradient buffers and the p.grad views of a bound reducer do not move between steps)."""
    if not (dsts[0].is_cuda and all(d.dtype == torch.float32 and s_.dtype == torch.float32 and d.is_contiguous() and s_.is_contiguous()
                                    for d, s_ in zip(dsts, srcs))):
        torch._foreach_add_(dsts, srcs)
        return
    key = tuple((d.data_ptr(), s_.data_ptr(), d.numel()) for d, s_ in zip(dsts, srcs))
    ent = _ADD_TABLES.get(key)
    if ent is None:
        rows, blk0 = [], 0
        for dptr, sptr, n in key:
            rows.append([dptr, sptr, n, 0, blk0])
            blk0 += -(-n // 1024)
        if len(_ADD_TABLES) > 64:
            _ADD_TABLES.clear()
        ent = _ADD_TABLES[key] = (torch.tensor(rows, dtype=torch.int64).to(dsts[0].device), len(rows), blk0)
    check(lib().yp_multi_add(ent[0].data_ptr(), ent[1], ent[2], _hip.stream_ptr()))


# modules whose parameters the keypoint / descriptor heads depend on (everything a forward whose Detect outputs take no part in the loss
# back-propagates through -- the warped pass of a training step, reference train.py:220-241); every other parameter only receives
# gradient from the full backward of the first pass.  TrainGraph._build marks the same split on its tape (checked by a GPU test).
KP_BRANCH_MODULES = ("Conv1", "Conv2", "Bottleneck1", "Conv3", "BottleneckDet", "ConvDet", "Bottleneck2", "ConvDescA", "ConvDescB", "BottleneckDesc",
                     "ConvDesc")


def grad_ready_groups(net):
    """Parameters of a YOLOPoint / YOLOPointv52 module in the order their gradients become FINAL in a training step that runs the
    full backward (image pass) before the keypoint-only backward (warped pass): first the parameters only the full pass reaches
    (Detect, PAN, YOLO encoder -- in reverse registration order, heads first), then the shared trunk and the keypoint / descriptor
    heads, which both passes contribute to.  Returns [("detector", [...]), ("keypoint", [...])]: the bucket plan of dp.GradAllReducer
    -- the detector buckets are all-reduced while the second backward pass is still running."""
    named = list(reversed(list(net.named_parameters())))
    # cv1 / cv2 of every C3 block side by side, tensor kind by tensor kind: laid out back to back (dp.GradAllReducer.flatten_parameters)
    # the two layers are one Conv with 2c_ output channels for the training plans (TrainGraph.merged_siblings)
    by_name = dict(named)
    for prefix, mod in net.named_modules():
        if type(mod).__name__ != "C3":
            continue
        six = [f"{prefix}.{cv}.{t}" for t in ("conv.weight", "bn.weight", "bn.bias") for cv in ("cv1", "cv2")]
        if not all(k in by_name and by_name[k].requires_grad for k in six):
            continue
        pos = min(i for i, (k, _) in enumerate(named) if k in six)
        named = [kv for kv in named if kv[0] not in six]
        named[pos:pos] = [(k, by_name[k]) for k in six]
    kp, det = [], []
    for name, p in named:
        if not p.requires_grad:
            continue
        (kp if name.split(".")[0] in KP_BRANCH_MODULES else det).append(p)
    return [("detector", det), ("keypoint", kp)]


def link_siblings(net):
    """BatchNorm running statistics of every C3 block's cv1 / cv2 back to back in memory (values kept; the buffers become views of one
    tensor) -- the buffer half of what TrainGraph.merged_siblings needs; the parameter half is the arena order of grad_ready_groups."""
    with torch.no_grad():
        for mod in net.modules():
            if type(mod).__name__ != "C3" or not hasattr(mod.cv1, "bn"):
                continue
            for name in ("running_mean", "running_var"):
                u, v = getattr(mod.cv1.bn, name), getattr(mod.cv2.bn, name)
                if u.data_ptr() + 4 * u.numel() == v.data_ptr() and u.untyped_storage().data_ptr() == v.untyped_storage().data_ptr():
                    continue
                both = torch.cat((u, v))
                u.data, v.data = both[:u.numel()], both[u.numel():]


def train_forward_pair(net, x, x_w):
    """Both train-mode forwards of a step (reference train.py:208,220: model(img), model(img_warp)) as one native pass.  Returns
    (outs, outs_w, heads, graph): the two output dicts of the reference's two calls (the warped pass's 'objects' are not produced: no loss
    reads them), the raw head tensors they are views of (semi / desc over 2B samples, the Detect levels) for torch.autograd.grad, and the
    TrainGraph for run_native_backward_pair.  BatchNorm batch statistics are per pass and the running statistics take the image pass's
    update first, exactly as two calls do."""
    params = list(net.parameters())
    heads = _YOLOPointPairFn.apply(net, x, x_w, *params)
    B = x.shape[0]
    semi, desc = heads[0], heads[1]
    if type(net).__name__ == "YOLOPointv52":
        desc = desc.div(torch.unsqueeze(torch.norm(desc, p=2, dim=1), 1))
    outs = {'semi': semi[:B], 'desc': desc[:B], 'objects': list(heads[2:]), 'semi_pair': semi, 'desc_pair': desc}
    outs_w = {'semi': semi[B:], 'desc': desc[B:], 'objects': None}
    return outs, outs_w, tuple(heads), heads[0].grad_fn.graph


def train_forward(net, x, with_graph=False):
    """Train-mode forward of a YOLOPoint module through the native plans, differentiable w.r.t. its parameters.
    with_graph: also return the raw head tensors of the native forward (semi, desc, the Detect levels) and the TrainGraph that
    produced them, for callers that drive the native backward themselves (engine.TrainStep: run_native_backward)."""
    params = list(net.parameters())
    outs = _YOLOPointTrainFn.apply(net, x, *params)
    desc = outs[1]
    if type(net).__name__ == "YOLOPointv52":          # reference models/YOLOPoint.py:318-319; YOLOPoint normalises inside the plan
        desc = desc.div(torch.unsqueeze(torch.norm(desc, p=2, dim=1), 1))
    res = {'semi': outs[0], 'desc': desc, 'objects': list(outs[2:])}
    if with_graph:
        return res, tuple(outs), outs[0].grad_fn.graph
    return res
