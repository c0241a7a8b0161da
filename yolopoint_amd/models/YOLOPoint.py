"""Model zoo + meta wrapper with the reference's API surface
(reference: src/models/YOLOPoint.py:17-145 Model, :148-246 YOLOPoint).

Same constructor arguments, same submodule attribute names (model.Conv1 ... model.ConvDesc,
model.Detect), same state_dict keys/shapes (fp32 OIHW in the file; repacked to NHWC-ordered
rows in the compute dtype when a plan is built), same forward() contract:
    {'semi': [B,65,H/8,W/8], 'desc': [B,c3,H/8,W/8], 'objects': list[3] (train) | (pred, list[3]) (eval)}
The forward itself is one native plan replay (csrc/plan.hip).
"""
import math
from copy import deepcopy
from ..switches import sw

import torch
from torch import nn

from .. import _hip
from ..plan import PlanBuilder, pack_input
from ..utils.general_yolo import make_divisible, LOGGER
from ..utils.torch_utils_yolo import fuse_conv_and_bn
from .common import Conv, C3, C2f, SPPF, HipModule
from .yolo import Detect

anchors_default = [
    [10, 13, 16, 30, 33, 23],
    [30, 61, 62, 45, 59, 119],
    [116, 90, 156, 198, 373, 326],
]

_VERSIONS = {'n': (0.33, 0.25), 's': (0.33, 0.5), 'm': (0.67, 0.75), 'l': (1., 1.), 'x': (1.33, 1.25)}


class YOLOPoint(HipModule):
    """Shared CSPDarknet encoder + YOLO PAN/Detect head + keypoint (semi) head + descriptor head."""
    plans_cover_all_reads = True           # every reader of its training buffers is a plan op (read through the class's own __dict__: subclasses must re-declare)

    def __init__(self, width_multiple=1., depth_multiple=1., inp_ch=3, nc=80, anchors=None):
        super().__init__()
        c1, c2, c3, c4, c5 = [make_divisible(2 ** k * width_multiple, 8) for k in range(6, 11)]
        n1, n2, n3 = [max(round(k * depth_multiple), 1) for k in (3, 6, 9)]
        # shared backbone
        self.Conv1 = Conv(inp_ch, c1, 6, 2, 2)
        self.Conv2 = Conv(c1, c2, 3, 2)
        self.Bottleneck1 = C3(c2, c2, n1)
        self.Conv3 = Conv(c2, c3, 3, 2)
        self.Bottleneck2 = C3(c3, c3, n2)
        # YOLO-exclusive backbone
        self.Conv4 = Conv(c3, c4, 3, 2)
        self.Bottleneck3 = C3(c4, c4, n3)
        self.Conv5 = Conv(c4, c5, 3, 2)
        self.Bottleneck4 = C3(c5, c5, n1)
        self.SPPooling = SPPF(c5, c5, 5)
        # object detector head (PAN)
        self.Conv6 = Conv(c5, c4, 1, 1, 0)
        self.Bottleneck5 = C3(c5, c4, n1)
        self.Conv7 = Conv(c4, c3, 1, 1, 0)
        self.Bottleneck6 = C3(c4, c3, n1)
        self.Conv8 = Conv(c3, c3, 3, 2, 1)
        self.Bottleneck7 = C3(c4, c4, n1)
        self.Conv9 = Conv(c4, c4, 3, 2, 1)
        self.Bottleneck8 = C3(c5, c5, n1)
        self.Detect = Detect(nc, anchors=anchors, ch=(c3, c4, c5))
        # keypoint detector head
        self.BottleneckDet = C3(c3, c3, n1)
        self.ConvDet = nn.Conv2d(c3, 65, 1, 1, 0, bias=False)
        # descriptor head
        self.ConvDescB = Conv(c3, c2, 3, 2, 1)
        self.ConvDescA = Conv(c2, c2, 3, 2, 1)
        self.ups = torch.nn.Upsample(scale_factor=(2, 2), mode='nearest')
        self.BottleneckDesc = C3(c3, c3, n1)
        self.ConvDesc = nn.Conv2d(c3, c3, 3, 1, 1, bias=False)
        self.static_outputs = False

    # ---------------------------------------------------------------------------------
    def _emit_stem(self, pb, img):
        """Conv1.  16-bit eval plans use the fused stem kernel, which reads the caller's NCHW fp32 image directly
        (no pack kernel, no NHWC image copy); its launch is handed back through pb.stem_launch."""
        c = self.Conv1.conv
        fusable = (pb.code != _hip.YP_F32 and c.kernel_size == (6, 6) and c.stride == (2, 2) and c.padding == (2, 2) and c.in_channels <= 4
                   and c.out_channels % 16 == 0 and c.out_channels <= 64 and isinstance(self.Conv1.act, nn.SiLU) and getattr(self, "fuse_stem", True)
                   and sw("YP_FUSE_STEM") != "0")
        pb.stem_launch = None
        if not fusable:
            pb.scope.append("Conv1")
            try:
                return self.Conv1.emit(pb, img)
            finally:
                pb.scope.pop()
        w, b = self.Conv1.folded()
        pb.scope.append("Conv1")
        out, pb.stem_launch = pb.stem(w, b, _hip.YP_ACT_SILU, img.H, img.W)
        pb.scope.pop()
        return out

    def _emit_stem_conv2(self, pb, img, run, block=None):
        """Conv1 + Conv2.  -s width (32 stem channels, 3x3 / stride-2 Conv2 of <= 64 channels), 16-bit eval plans: ONE launch whose stem output
        stays in LDS (PlanBuilder.stem_conv2; YP_FUSE_STEM2=0: the two launches)."""
        import os
        c1, c2 = self.Conv1.conv, self.Conv2.conv
        ok = (pb.code != _hip.YP_F32 and c1.kernel_size == (6, 6) and c1.stride == (2, 2) and c1.padding == (2, 2) and c1.in_channels <= 4
              and c1.out_channels == 32 and isinstance(self.Conv1.act, nn.SiLU) and getattr(self, "fuse_stem", True)
              and c2.kernel_size == (3, 3) and c2.stride == (2, 2) and c2.padding == (1, 1) and c2.in_channels == 32 and c2.out_channels <= 64
              and c2.out_channels % 8 == 0 and isinstance(self.Conv2.act, nn.SiLU) and img.H % 4 == 0 and img.W % 4 == 0
              # (the never-materialised stem output is addressed with 32-bit byte offsets: B (H/2) (W/2) 32 channels x 2 bytes < 2^31 -- beyond
              # that, ~328 images of 640 x 640, the library refuses the fused launch and the two-launch plan, which has no such limit, is taken)
              and pb.B * (img.H // 2) * (img.W // 2) * 32 * 2 + 32 * 2 + 64 < (1 << 31)
              and sw("YP_FUSE_STEM") != "0" and sw("YP_FUSE_STEM2") != "0")
        if not ok:
            out = run("Conv2", self.Conv2, self._emit_stem(pb, img))
            return (out, False) if block is not None else out
        (w1, b1), (w2, b2) = self.Conv1.folded(), self.Conv2.folded()
        # ... and Bottleneck1's cv1 + cv2 (a C3 of hidden width 32 over Conv2's 64 channels: one 64 -> 64 pointwise filter) in the same
        # launch: Conv2's output feeds nothing else, so it is never written (YP_FUSE_STEM3=0: two launches).  Returns the block's output.
        from .common import C3
        b1m = self.Bottleneck1
        if (block is not None and isinstance(b1m, C3) and c2.out_channels == 64 and b1m.cv1.conv.in_channels == 64 and b1m.cv1.conv.out_channels == 32
                and isinstance(b1m.cv1.act, nn.SiLU) and isinstance(b1m.cv2.act, nn.SiLU) and sw("YP_FUSE_STEM3") != "0"):
            w12, b12 = b1m.merged_cv12()
            H2, W2 = img.H // 4, img.W // 4
            cat = pb.new_buf(H2, W2, 64)
            t = pb.new_buf(H2, W2, 32).view()
            pb.scope.append("Conv1+Conv2+Bottleneck1.cv1+cv2")
            _, pb.stem_launch = pb.stem_conv2(w1, b1, _hip.YP_ACT_SILU, img.H, img.W, w2, b2, _hip.YP_ACT_SILU,
                                              post=(w12, b12, _hip.YP_ACT_SILU, t, cat.view(32, 32)))
            pb.scope.pop()
            return block("Bottleneck1", b1m, None, pre=(t, cat)), True
        from .common import C2f
        if (block is not None and isinstance(b1m, C2f) and c2.out_channels == 64 and b1m.cv1.conv.in_channels == 64 and b1m.cv1.conv.out_channels == 64
                and b1m.cv1.conv.kernel_size == (1, 1) and isinstance(b1m.cv1.act, nn.SiLU) and sw("YP_FUSE_STEM3") != "0"):
            wc, bc = b1m.cv1.folded()                   # (v52: the C2f's cv1 fills channels [0, 2c) of its concat buffer)
            cat = pb.new_buf(img.H // 4, img.W // 4, (2 + len(b1m.m)) * b1m.c)
            pb.scope.append("Conv1+Conv2+Bottleneck1.cv1")
            _, pb.stem_launch = pb.stem_conv2(w1, b1, _hip.YP_ACT_SILU, img.H, img.W, w2, b2, _hip.YP_ACT_SILU,
                                              post=(wc, bc, _hip.YP_ACT_SILU, cat.view(0, 64), None))
            pb.scope.pop()
            return block("Bottleneck1", b1m, None, pre=cat), True
        pb.scope.append("Conv1+Conv2")
        out, pb.stem_launch = pb.stem_conv2(w1, b1, _hip.YP_ACT_SILU, img.H, img.W, w2, b2, _hip.YP_ACT_SILU)
        pb.scope.pop()
        return (out, False) if block is not None else out

    def emit(self, pb, img, decode=True):
        """Dataflow of reference models/YOLOPoint.py:198-246; cat/ups are views, never copies."""
        def run(name, mod, x, **kw):
            pb.scope.append(name)
            try:
                return mod.emit(pb, x, **kw)
            finally:
                pb.scope.pop()

        # Schedule (two lanes, PlanBuilder.side).  The keypoint head, the descriptor head and the first two Detect levels feed nothing but the
        # caller, and the YOLO encoder / PAN chain behind Bottleneck3 is 30 dependent launches of 100-400 workgroups on 256 CUs: the heads
        # (P3-sized launches that fill the chip) go to the plan's side lane and are forked behind Bottleneck4 -- legal anywhere behind
        # Bottleneck2, they read only xa / x8 / xb -- so they run beside that chain instead of in front of it.  Measured (batch 8, 640x640,
        # f16, same box): one lane 0.751 ms, heads forked behind Bottleneck2 / 3 / 4: 0.706 / 0.706 / 0.691 ms (eager two-stream replay).
        x, fused_b1 = self._emit_stem_conv2(pb, img, run, block=run)
        xa = x if fused_b1 else run("Bottleneck1", self.Bottleneck1, x)
        x8 = run("Conv3", self.Conv3, xa)
        xb = run("Bottleneck2", self.Bottleneck2, x8)
        fork = int(sw("YP_HEADS_FORK"))        # the heads are forked behind Bottleneck<fork> (2 / 3 / 4; measurements above)
        heads = {}

        def emit_heads():
            with pb.side():
                # keypoint head
                t = run("BottleneckDet", self.BottleneckDet, x8)
                pb.scope.append("ConvDet")
                heads["semi"] = pb.conv(t, self.ConvDet.weight.detach().float(), None, 1, 1, 0, _hip.YP_ACT_NONE, out_f32=True)
                pb.scope.pop()
                # descriptor head
                dA = run("ConvDescA", self.ConvDescA, xa)
                dB = run("ConvDescB", self.ConvDescB, xb)
                d = run("BottleneckDesc", self.BottleneckDesc, [dA, dB.up()])
                pb.scope.append("ConvDesc")
                heads["desc"] = pb.conv(d, self.ConvDesc.weight.detach().float(), None, 3, 1, 1, _hip.YP_ACT_NONE, out_f32=True)
                pb.l2norm(heads["desc"], heads["desc"], self.ConvDesc.out_channels)
                pb.scope.pop()
                self._emit_heads_hook(pb)
        if fork <= 2:
            emit_heads()
        # YOLO encoder
        x = run("Conv4", self.Conv4, xb)
        xc = run("Bottleneck3", self.Bottleneck3, x)
        if fork == 3:
            emit_heads()
        x = run("Conv5", self.Conv5, xc)
        x = run("Bottleneck4", self.Bottleneck4, x)
        if fork >= 4:
            emit_heads()
        semi, desc = heads["semi"], heads["desc"]
        x = run("SPPooling", self.SPPooling, x)
        # PAN head; every Detect level right behind the block it reads
        pb.scope.append("Detect")
        det = self.Detect.emit_begin(pb, [(xb.LH, xb.LW), (xb.LH // 2, xb.LW // 2), (xb.LH // 4, xb.LW // 4)], decode)
        pb.scope.pop()

        def detect_level(i, v, side):
            pb.scope.append("Detect")
            with pb.side(side):
                self.Detect.emit_level(pb, det, i, v)
            pb.scope.pop()
        xd = run("Conv6", self.Conv6, x)
        x = run("Bottleneck5", self.Bottleneck5, [xd.up(), xc])
        xe = run("Conv7", self.Conv7, x)
        xf = run("Bottleneck6", self.Bottleneck6, [xe.up(), xb])
        detect_level(0, xf, True)
        x = run("Conv8", self.Conv8, xf)
        xg = run("Bottleneck7", self.Bottleneck7, [x, xe])
        detect_level(1, xg, True)
        x = run("Conv9", self.Conv9, xg)
        p5 = run("Bottleneck8", self.Bottleneck8, [x, xd])
        detect_level(2, p5, False)
        z, xs = det["z"], det["outs"]
        return {"semi": semi, "desc": desc, "z": z, "xs": xs}

    def _emit_heads_hook(self, pb):
        """`heads_hook = True` (set by frontend.YoloPointFrontend): a callback op behind the keypoint / descriptor heads on the side lane.  At
        replay it calls `self._heads_cb(stream)` if one is set -- the front end enqueues the keypoint decode / NMS there, so that the post-
        processing that needs `semi` only runs beside the YOLO encoder / PAN chain instead of behind the whole forward."""
        if getattr(self, "heads_hook", False):
            pb.callback(lambda stream: (self.__dict__.get("_heads_cb") or (lambda s_: None))(stream), "heads_hook")

    def build_plan(self, B, H, W, device, graph=False):
        """Build (and cache) the native plan for a [B, inp_ch, H, W] input."""
        if H % 32 or W % 32:
            raise _hip.YpError(f"input size {H}x{W} must be a multiple of the max stride 32")
        if self.Detect.stride is None:
            raise _hip.YpError("Detect.stride is not set (construct through models.Model)")
        code = _hip.dtype_code(self.compute_dtype)
        key = (B, H, W, code, self.training, self._weights_version(), torch.device(device).index, bool(graph), bool(getattr(self, "heads_hook", False)))
        cache = self.__dict__.setdefault("_plans", {})
        if key not in cache:
            cache.clear()
            pb = PlanBuilder(B, code, device)
            img = pb.new_buf(H, W, 4)
            outs = self.emit(pb, img.view(), decode=not self.training)
            plan = pb.finish()
            plan.stem_launch = pb.stem_launch
            plan.stem_record = getattr(pb, "stem_record", None) if pb.stem_launch else None
            # A plan with schedule lanes replays EAGERLY on its two streams even when a graph was asked for: captured into a hipGraph the
            # side branch bought 0-4 % (0.743 vs 0.752 ms at batch 8), the same launches on two plain streams 8 % (0.691 ms) -- the ~50
            # launches of a forward cost the host ~0.2 ms, well inside the step.  YP_LANES_EAGER=0: capture them.
            if graph and not plan.has_callbacks and not (plan.has_lanes and sw("YP_LANES_EAGER") != "0"):
                plan.instantiate_graph()
            cache[key] = (plan, img, outs)
        return cache[key]

    def run_plan(self, plan, img, x):
        """Input hand-over + plan replay: the fused stem reads x directly, otherwise x is packed to NHWC first."""
        if plan.stem_launch is not None:
            plan.stem_launch(x)
        else:
            pack_input(x, img.view(), plan.code)
        plan.run()

    def _train_graph(self, x, pair=False, fp8=False):
        """A free TrainGraph (static forward + backward plans and all their buffers) for this input shape.  The two
        forwards of one training step (image, warped image: reference train.py:208,220) get two graphs."""
        from ..training import TrainGraph
        code = _hip.dtype_code(self.compute_dtype)
        key = (tuple(x.shape), code, x.device.index, tuple(p.data_ptr() for p in self.parameters()), bool(pair), bool(fp8))
        pool = self.__dict__.setdefault("_train_graphs", {})
        graphs = pool.setdefault(key, [])
        for g in graphs:
            if not g.busy:
                return g
        if len(graphs) >= 4:
            raise _hip.YpError("more than 4 un-backpropagated train-mode forwards in flight for one input shape")
        g = TrainGraph(self, x.shape[0], x.shape[2], x.shape[3], code, x.device, pair=pair, fp8=fp8)
        graphs.append(g)
        return g

    @_hip.guarded
    def forward_with_graph(self, x):
        """Train-mode forward that also hands back the raw head tensors and the native graph (see training.train_forward)."""
        if not self.training:
            raise _hip.YpError("forward_with_graph is a train-mode entry point")
        from ..training import train_forward
        return train_forward(self, x.contiguous().float(), with_graph=True)

    @_hip.guarded
    def forward_pair(self, x, x_w):
        """model(img) and model(img_warp) of one training step (reference train.py:208,220) as ONE native pass over 2B samples with
        per-pass BatchNorm statistics (training.train_forward_pair)."""
        if not self.training:
            raise _hip.YpError("forward_pair is a train-mode entry point")
        if tuple(x.shape) != tuple(x_w.shape):
            raise _hip.YpError("forward_pair: the image and the warped image batch must have the same shape")
        from ..training import train_forward_pair
        return train_forward_pair(self, x.contiguous().float(), x_w.contiguous().float())

    @_hip.guarded
    def forward(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise _hip.YpError("YOLOPoint.forward needs a cuda (HIP) tensor: the hot path has no CPU fallback")
        if self.training:
            if any(not hasattr(m, "bn") for m in self.modules() if isinstance(m, Conv)):
                raise _hip.YpError("a fused model (Model.fuse()) cannot run in train mode")
            from ..training import train_forward
            return train_forward(self, x.contiguous().float())
        x = x.contiguous().float()
        B, C_, H, W = x.shape
        if C_ > 4:
            raise _hip.YpError("inp_ch > 4 is not supported by the stem kernel")
        plan, img, outs = self.build_plan(B, H, W, x.device, graph=getattr(self, "use_graph", False))
        self.run_plan(plan, img, x)
        semi = outs["semi"].buf.t[..., :65].permute(0, 3, 1, 2)
        dch = getattr(self, "_desc_channels", None) or self.ConvDesc.out_channels
        desc = outs["desc"].buf.t[..., :dch].permute(0, 3, 1, 2)
        xs = list(outs["xs"])
        z = outs["z"]
        if not self.static_outputs:
            semi, desc, xs = semi.clone(), desc.clone(), [t.clone() for t in xs]
            z = z.clone() if z is not None else None
        objects = xs if self.training else (z, xs)
        return {'semi': semi, 'desc': desc, 'objects': objects}


class YOLOPointv52(YOLOPoint):
    """C2f variant (reference: models/YOLOPoint.py:248-342): no Conv6/Conv7, a 65-channel C2f as keypoint head, MaxPool2d(2,2)
    on the stride-4 features for the descriptor branch.  Inference (eval) only in this build."""
    plans_cover_all_reads = False          # (its descriptor normalisation is differentiated in PyTorch, outside the plans' access lists)

    def __init__(self, width_multiple=1., depth_multiple=1., inp_ch=3, nc=80, anchors=None):
        HipModule.__init__(self)
        c1, c2, c3, c4, c5 = [make_divisible(2 ** k * width_multiple, 8) for k in range(6, 11)]
        n1, n2, n3 = [max(round(k * depth_multiple), 1) for k in (3, 6, 9)]
        self.Conv1 = Conv(inp_ch, c1, 6, 2, 2)
        self.Conv2 = Conv(c1, c2, 3, 2)
        self.Bottleneck1 = C2f(c2, c2, n1)
        self.Conv3 = Conv(c2, c3, 3, 2)
        self.Bottleneck2 = C2f(c3, c3, n2)
        self.Conv4 = Conv(c3, c4, 3, 2)
        self.Bottleneck3 = C2f(c4, c4, n3)
        self.Conv5 = Conv(c4, c4, 3, 2)
        self.Bottleneck4 = C2f(c4, c4, n1)
        self.SPPooling = SPPF(c4, c4, 5)
        self.Bottleneck5 = C2f(c5, c4, n1)
        self.Bottleneck6 = C2f(c4 + c3, c3, n1)
        self.Conv8 = Conv(c3, c3, 3, 2, 1)
        self.Bottleneck7 = C2f(c4 + c3, c4, n1)
        self.Conv9 = Conv(c4, c4, 3, 2, 1)
        self.Bottleneck8 = C2f(c5, c4, n1)
        self.Detect = Detect(nc, anchors=anchors, ch=(c3, c4, c4))
        self.BottleneckDet = C2f(c3, 65, n1)
        self.ConvDescB = Conv(c3, c2, 3, 2, 1)
        self.MaxPool = torch.nn.MaxPool2d(kernel_size=2, stride=2)
        self.ups = torch.nn.Upsample(scale_factor=(2, 2), mode='nearest')
        self.BottleneckDesc = C2f(c3, c3, n1)
        self.static_outputs = False
        self._desc_channels = c3

    def emit(self, pb, img, decode=True):
        """Dataflow of reference models/YOLOPoint.py:294-342."""
        def run(name, mod, x, **kw):
            pb.scope.append(name)
            try:
                return mod.emit(pb, x, **kw)
            finally:
                pb.scope.pop()

        x, fused_b1 = self._emit_stem_conv2(pb, img, run, block=run)
        xa = x if fused_b1 else run("Bottleneck1", self.Bottleneck1, x)
        x8 = run("Conv3", self.Conv3, xa)
        xb = run("Bottleneck2", self.Bottleneck2, x8)
        x = run("Conv4", self.Conv4, xb)
        xc = run("Bottleneck3", self.Bottleneck3, x)
        x = run("Conv5", self.Conv5, xc)
        x = run("Bottleneck4", self.Bottleneck4, x)
        with pb.side():                       # (heads and the first two Detect levels on the side lane: see YOLOPoint.emit)
            semi = run("BottleneckDet", self.BottleneckDet, x8, out_f32=True)
            dA = pb.new_buf(xa.LH // 2, xa.LW // 2, xa.C).view()
            pb.scope.append("MaxPool")
            pb.op(_hip.OP_MAXPOOL2, [xa], [dA], "", v=[xa, dA], i=[pb.code, pb.B])
            pb.scope.pop()
            dB = run("ConvDescB", self.ConvDescB, xb)
            desc = run("BottleneckDesc", self.BottleneckDesc, [dA, dB.up()], out_f32=True)
            pb.scope.append("BottleneckDesc")
            pb.l2norm(desc, desc, self._desc_channels)
            pb.scope.pop()
            self._emit_heads_hook(pb)
        xd = run("SPPooling", self.SPPooling, x)
        pb.scope.append("Detect")
        det = self.Detect.emit_begin(pb, [(xb.LH, xb.LW), (xb.LH // 2, xb.LW // 2), (xb.LH // 4, xb.LW // 4)], decode)
        pb.scope.pop()

        def detect_level(i, v, side):
            pb.scope.append("Detect")
            with pb.side(side):
                self.Detect.emit_level(pb, det, i, v)
            pb.scope.pop()
        xe = run("Bottleneck5", self.Bottleneck5, [xd.up(), xc])
        xf = run("Bottleneck6", self.Bottleneck6, [xe.up(), xb])
        detect_level(0, xf, True)
        x = run("Conv8", self.Conv8, xf)
        xg = run("Bottleneck7", self.Bottleneck7, [x, xe])
        detect_level(1, xg, True)
        x = run("Conv9", self.Conv9, xg)
        p5 = run("Bottleneck8", self.Bottleneck8, [x, xd])
        detect_level(2, p5, False)
        z, xs = det["z"], det["outs"]
        return {"semi": semi, "desc": desc, "z": z, "xs": xs}



class Model(nn.Module):
    """Meta wrapper (reference: models/YOLOPoint.py:17-145)."""

    def __init__(self, names=(), model_name='YOLOPoint', version=None, inp_ch=3, anchors=None):
        super().__init__()
        anchors = anchors or anchors_default
        nc = len(names) if hasattr(names, '__len__') and len(names) > 0 else 1
        version = version.lower() if isinstance(version, str) else version
        if version is None:
            dm, wm = None, None
        elif version in _VERSIONS:
            dm, wm = _VERSIONS[version]
        else:
            raise Exception(f'Version {version} is not a valid input. Choose one of n, s, m, l, x.')
        from ..utils.utils import load_model
        self.model = load_model(meta_model=False, width_multiple=wm, depth_multiple=dm, inp_ch=inp_ch, nc=nc,
                                anchors=anchors, model_name=model_name)
        if hasattr(self.model, 'Detect'):
            m = self.model.Detect
            # The reference measures the strides with a dummy 256x256 forward (YOLOPoint.py:61-65); the
            # three detection levels sit after 3, 4 and 5 stride-2 convolutions, i.e. 8 / 16 / 32.
            m.stride = torch.tensor([8., 16., 32.])
            m.anchors /= m.stride.view(-1, 1, 1)
            self._check_anchor_order(m)
            self._initialize_biases()

    @staticmethod
    def _check_anchor_order(m):
        """Anchor areas must grow with the strides (reference YOLOPoint.py:20-28): a list given large-to-small is turned around."""
        areas = m.anchors.prod(-1).view(-1)
        direction = lambda first, last: (float(last) > float(first)) - (float(last) < float(first))      # -1 / 0 / +1
        if direction(areas[0], areas[-1]) != direction(m.stride[0], m.stride[-1]):
            LOGGER.info('Reversing anchor order')
            m.anchors.copy_(torch.flip(m.anchors, dims=(0,)))

    def forward(self, x):
        return self.model(x)

    def _apply(self, fn):
        """.to() / .cuda() / .float() also move the Detect head's plain-tensor attributes (stride and the cached decode grids, which
        are not registered buffers; reference YOLOPoint.py:73-82)."""
        moved = super()._apply(fn)
        head = getattr(moved.model, 'Detect', None)
        if head is not None:
            head.stride = fn(head.stride)
            head.grid = [fn(g) for g in head.grid]
            if isinstance(head.anchor_grid, list):
                head.anchor_grid = [fn(g) for g in head.anchor_grid]
        return moved

    # -- compute precision ----------------------------------------------------------------
    def set_compute_dtype(self, dt):
        """'f16' | 'bf16' | 'f32': arithmetic type of activations and packed weights (fp32 accumulate)."""
        _hip.dtype_code(dt)
        for m in self.modules():
            if isinstance(m, HipModule):
                m.compute_dtype = dt
        return self

    def half(self):       # parameters stay fp32 masters; .half() selects the f16 compute path
        return self.set_compute_dtype("f16")

    def bfloat16(self):
        return self.set_compute_dtype("bf16")

    def float(self):
        return self.set_compute_dtype("f32")

    def fuse(self):
        """Fold every Conv's BatchNorm into its Conv2d (reference: YOLOPoint.py:84-90)."""
        for m in self.model.modules():
            if isinstance(m, Conv) and hasattr(m, 'bn'):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, 'bn')
        return self

    def _initialize_biases(self, cf=None):
        """Detect bias prior (reference YOLOPoint.py:92-100): objectness as if 8 objects fell on a 640-px image at the level's
        stride, classes as if each had prior 0.6 / nc (or the given class frequencies)."""
        head = self.model.Detect
        class_prior = math.log(0.6 / (head.nc - 0.999999)) if cf is None else torch.log(cf / cf.sum())
        for conv, stride in zip(head.m, head.stride):
            with torch.no_grad():
                per_anchor = conv.bias.detach().clone().view(head.na, -1)
                per_anchor[:, 4] += math.log(8 / (640 / float(stride)) ** 2)
                per_anchor[:, 5:] += class_prior
            conv.bias = torch.nn.Parameter(per_anchor.reshape(-1), requires_grad=True)

    def load_state_dict(self, target_state_dict, strict=True, verbose=False):
        """Checkpoints of either prefix style ('model.Conv1...' or 'Conv1...') load; one whose Detect head was built for another
        class count keeps everything that still fits and leaves the head freshly initialised (reference YOLOPoint.py:102-120)."""
        from .common import invalidate_packed_weights
        probe = next((k for k in ('model.Detect.m.0.bias', 'Detect.m.0.bias') if k in target_state_dict), None)
        if probe is None:                       # a checkpoint of the bare network, or of a model without a Detect head
            try:
                self.model.load_state_dict(target_state_dict, strict=strict)
            except RuntimeError:
                super().load_state_dict(target_state_dict, strict=strict)
        elif tuple(target_state_dict[probe].shape) == tuple(self.state_dict()[probe].shape):
            super().load_state_dict(target_state_dict, strict)
        else:
            if verbose:
                LOGGER.info("Number of classes have changed. Reinitializing Detect layer.\n")
            self.load_partial_state_dict(target_state_dict, strict, verbose)
        invalidate_packed_weights()

    def load_partial_state_dict(self, target_state_dict, strict=True, verbose=False):
        """Walk both key lists in step; where the trailing '<module>.<tensor>' name and the shape agree, take the checkpoint's tensor
        (reference YOLOPoint.py:122-135, including its pairing of the two key lists by position)."""
        own = self.state_dict()
        merged = deepcopy(own)
        tail = lambda key: key.rsplit('.', 2)[-2:]
        for mine, theirs in zip(list(own), list(target_state_dict)):
            if tail(mine) != tail(theirs) or own[mine].shape != target_state_dict[theirs].shape:
                continue
            if verbose:
                LOGGER.info(f"{mine} {' ' * (50 - len(mine))} {theirs}")
            merged[theirs] = target_state_dict[mine]
        super().load_state_dict(merged, strict)

    def freeze_layers(self, to_freeze, verbose=True):
        """requires_grad = False for the parameters whose position in named_parameters() is listed (reference YOLOPoint.py:137-145)."""
        wanted = set(int(i) for i in to_freeze)
        if verbose:
            LOGGER.info("Freezing weights...")
        for pos, (name, param) in enumerate(self.named_parameters()):
            hit = pos in wanted
            if verbose:
                LOGGER.info(f"{pos} {name} {' ' * (45 - len(name) - len(str(pos)))} {'--> freeze' if hit else ''}")
            if hit:
                param.requires_grad_(False)
