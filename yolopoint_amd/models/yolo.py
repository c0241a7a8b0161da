"""YOLOv5 detection head (reference: src/models/yolo.py:34-91).

`m[i]` are nn.Conv2d parameter containers (1x1, bias) exactly like the reference, `anchors` is
the same registered buffer (grid units), `stride` the same plain attribute.  The 1x1 convs run on
the implicit-GEMM kernel with fp32 output; view/permute/sigmoid/grid decode/concat
(yolo.py:53-68) are one fused kernel per level (csrc/elementwise.hip: detect_decode_kernel)
that writes the permuted raw tensor and its rows of the [B, 25200, 85] prediction directly.
"""
import torch
import torch.nn as nn

from .. import _hip
from ..plan import PlanBuilder, pack_input, round_up
from .common import HipModule


class Detect(HipModule):
    stride = None  # strides computed during build

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        self.anchor_grid = [torch.zeros(1)] * self.nl
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace

    def emit_begin(self, pb, dims, decode):
        """Allocate the outputs of an nl-level head over feature maps of sizes dims = [(ny, nx)]; the levels are then emitted one by one
        (emit_level) wherever their inputs become available in the launch list."""
        assert len(dims) == self.nl
        rows = [self.na * ny * nx for ny, nx in dims]
        z = pb.new_tensor((pb.B, sum(rows), self.no)) if decode else None
        return dict(z=z, rows=rows, total=sum(rows), outs=[None] * self.nl)

    def emit_level(self, pb, st, i, v):
        pb.scope.append(f"m.{i}")
        ny, nx = v.LH, v.LW
        assert self.na * ny * nx == st["rows"][i]
        xo = pb.new_tensor((pb.B, self.na, ny, nx, self.no))
        stride = float(self.stride[i])
        anchors_px = (self.anchors[i].detach().float().cpu() * stride).reshape(-1).tolist()
        # 1x1 conv + (view/permute/sigmoid/grid decode/concat) in ONE launch: the raw logits never round-trip HBM
        pb.conv(v, self.m[i].weight.detach().float(), self.m[i].bias.detach().float(), 1, 1, 0, _hip.YP_ACT_NONE, out_f32=True,
                detect=dict(na=self.na, no=self.no, stride=stride, anchors_px=anchors_px, x_out=xo, z_out=st["z"],
                            rows_total=st["total"], row_offset=sum(st["rows"][:i])))
        pb.scope.pop()
        st["outs"][i] = xo

    def emit(self, pb, xs, decode):
        """xs: list of nl feature Views.  Returns (z or None, [x_i])."""
        st = self.emit_begin(pb, [(v.LH, v.LW) for v in xs], decode)
        for i, v in enumerate(xs):
            self.emit_level(pb, st, i, v)
        return st["z"], st["outs"]

    def forward(self, x):
        """x: list of nl NCHW cuda tensors (like the reference, which it mutates in place)."""
        if not all(t.is_cuda for t in x):
            raise _hip.YpError("Detect.forward needs cuda (HIP) tensors: the hot path has no CPU fallback")
        if self.stride is None:
            raise _hip.YpError("Detect.stride is not set (build the head through models.Model)")
        xs = [t.contiguous().float() for t in x]
        key = (tuple(tuple(t.shape) for t in xs), _hip.dtype_code(self.compute_dtype), self.training, self._weights_version())
        cache = self.__dict__.setdefault("_plans", {})
        if key not in cache:
            cache.clear()
            code = _hip.dtype_code(self.compute_dtype)
            pb = PlanBuilder(xs[0].shape[0], code, xs[0].device)
            inbs = [pb.new_buf(t.shape[2], t.shape[3], round_up(t.shape[1], pb.ce)) for t in xs]
            z, outs = self.emit(pb, [b.view() for b in inbs], decode=not self.training)
            cache[key] = (pb.finish(), inbs, z, outs)
        plan, inbs, z, outs = cache[key]
        for t, b in zip(xs, inbs):
            pack_input(t, b.view(), plan.code)
        plan.run()
        outs = [o.clone() for o in outs]
        for i in range(self.nl):
            x[i] = outs[i]
        return x if self.training else (z.clone(), x)
