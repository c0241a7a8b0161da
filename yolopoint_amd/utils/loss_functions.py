"""Training losses of the YOLOPoint step, in PyTorch autograd (they are small, data-dependent
gather/scatter work between the native forward and backward) — same call signatures and values as the
reference: src/utils/loss_functions.py:90-234 (ComputeObjectLoss), :484-597 (infonce), :600-619
(ComputeDetectorLoss); train.py:212-241 combines them.

The random draws of `infonce` (match shuffling, negative sampling) can be injected (`perm_fn`, `randint_fn`)
so that the loss is reproducible in tests; by default they are the reference's torch.randperm / np.random.randint.
"""
import os
from ..switches import sw

import numpy as np
import torch
import torch.nn.functional as F

from .._hip import guarded as _guarded
from .torch_utils_yolo import de_parallel
from .utils import getMasks


_CONST = {}


def _const(key, make):
    """Small shape-dependent constants (grids, offsets) live on the device once instead of being rebuilt on the host and
    copied (a synchronising pageable H2D copy) in every training step."""
    if key not in _CONST:
        _CONST[key] = make()
    return _CONST[key]


class ComputeDetectorLoss:
    """Keypoint-detector loss: BCE between softmax(semi) and the 65-channel cell labels, summed over channels,
    averaged over valid cells (reference :600-619)."""

    def __init__(self, device):
        self.device = device

    def __call__(self, inp, target, mask, groups=1):
        """groups > 1 (device path): `inp` / `target` / `mask` hold `groups` consecutive sample sets and the result is the SUM of the
        losses of the sets (train.py:232: loss_det + loss_det_warp with both passes' logits in one tensor) -- one gradient tensor for
        the whole input instead of one zero-padded tensor per set."""
        if (inp.is_cuda and inp.dtype == torch.float32 and inp.dim() == 4 and inp.shape[1] == 65 and target.dtype == torch.float32
                and sw("YP_NATIVE_DETLOSS") != "0"):
            return _DetLossNative.apply(inp, target, mask.float().contiguous(), groups)
        if groups != 1:
            n = inp.shape[0] // groups
            return sum(self(inp[g * n:(g + 1) * n], target[g * n:(g + 1) * n], mask[g * n:(g + 1) * n]) for g in range(groups))
        per_cell = F.binary_cross_entropy(torch.softmax(inp, dim=1), target, reduction='none').sum(dim=1)
        return (per_cell * mask).sum() / (mask.sum() + 1e-10)


class _DetLossNative(torch.autograd.Function):
    """ComputeDetectorLoss through csrc/losses.hip (yp_detloss): softmax + BCE + mask + both reductions and the gradient w.r.t. the
    logits in two launches (one wavefront per cell), instead of ~10 PyTorch kernels forward and backward.  YP_NATIVE_DETLOSS=0 selects
    the PyTorch formulation above (also used for CPU tensors)."""

    @staticmethod
    @_guarded
    def forward(ctx, inp, target, mask, groups=1):
        from .. import _hip
        import ctypes as C
        B, _, Hc, Wc = inp.shape
        l = _hip.lib()
        dz = torch.empty_like(inp)                      # (same strides as inp)
        if dz.stride() != inp.stride():
            inp = inp.contiguous()
            dz = torch.empty_like(inp)
        target = target if target.shape[0] == B else target.contiguous()
        sums = torch.empty((groups, 2), dtype=torch.float32, device=inp.device)
        Bg = B // groups
        nb = l.yp_detloss_workspace_bytes(Bg, Hc, Wc)
        ws = torch.empty(nb, dtype=torch.uint8, device=inp.device)
        zs, ts = (C.c_int64 * 4)(*inp.stride()), (C.c_int64 * 4)(*target.stride())
        for g in range(groups):                         # (one launch pair per sample set: each has its own mask normalisation)
            _hip.check(l.yp_detloss(inp.data_ptr() + 4 * g * Bg * inp.stride(0), zs, target.data_ptr() + 4 * g * Bg * target.stride(0), ts,
                                    mask.data_ptr() + 4 * g * Bg * Hc * Wc, Bg, Hc, Wc, dz.data_ptr() + 4 * g * Bg * dz.stride(0), sums[g].data_ptr(),
                                    ws.data_ptr(), nb, _hip.stream_ptr()))
        inv = 1.0 / (sums[:, 1] + 1e-10)                # [groups]
        ctx.save_for_backward(dz, inv)
        ctx.groups = groups
        return (sums[:, 0] * inv).sum() if groups > 1 else sums[0, 0] * inv[0]

    @staticmethod
    @_guarded
    def backward(ctx, g):
        dz, inv = ctx.saved_tensors
        if ctx.groups == 1:
            return dz * (g * inv[0]), None, None, None
        per_sample = (g * inv).repeat_interleave(dz.shape[0] // ctx.groups).view(-1, 1, 1, 1)
        return dz * per_sample, None, None, None


class ComputeObjectLoss:
    """YOLOv5 box (CIoU) + objectness + class loss over the Detect levels -- the reference's `ComputeObjectLoss`
    (src/utils/loss_functions.py:90-234): same constructor, same call, same (loss [1], (box, obj, cls) [3]) result.

    Everything runs on the device: `assign()` = the reference's build_targets as one scan kernel (yp_build_targets: no boolean-mask
    indexing, no host synchronisation), `__call__` = three launches per level that produce the value AND the gradient
    (yp_objloss_level_dev).  There is no CPU path (the PyTorch statement of this loss lives with the test infrastructure, where the
    parity tests pin both to the reference's values and gradients)."""
    sort_obj_iou = False

    def __init__(self, model, config, device, autobalance=False):
        if config.get('fl_gamma', 0.0) > 0:
            raise NotImplementedError("focal loss (fl_gamma > 0) is not used by the reference configs")
        if autobalance:
            raise NotImplementedError("autobalance is off in every reference config (train.py:166-167)")
        head = de_parallel(model).model.Detect
        self.hyp, self.device, self.autobalance, self.gr = config, device, False, 1.0
        smoothing = config.get('label_smoothing', 0.0)
        self.cp, self.cn = 1.0 - 0.5 * smoothing, 0.5 * smoothing                  # smoothed BCE targets (positive, negative)
        self.balance = [4.0, 1.0, 0.4] if head.nl == 3 else [4.0, 1.0, 0.25, 0.06, 0.02]
        self.ssi = 0
        self.na, self.nc, self.nl, self.anchors = head.na, head.nc, head.nl, head.anchors

    # -- target assignment ------------------------------------------------------------------------------------------------
    @_guarded
    def assign(self, p, targets):
        """Entry lists of every level as device arrays with device-side counts (no host synchronisation).  `p`: the level
        tensors or just their shapes [(B, na, ny, nx, no)]."""
        from .. import _hip
        shapes = [tuple(t.shape) if isinstance(t, torch.Tensor) else tuple(t) for t in p]
        dev = targets.device if targets.is_cuda else torch.device(self.device)
        if dev.type != "cuda":
            raise _hip.YpError("ComputeObjectLoss runs on the device only (its CPU statement is test infrastructure, not product)")
        targets = targets.to(dev, torch.float32).contiguous()
        nt, nl, na = targets.shape[0], self.nl, self.na
        cap = max(5 * na * nt, 1)
        geo = _const(("objgeo", tuple(shapes), str(dev)), lambda: torch.tensor([[s[2], s[3]] for s in shapes], dtype=torch.int32, device=dev))
        anchors = self.anchors.to(dev, torch.float32).contiguous()
        ents = dict(cell=torch.empty((nl, cap), dtype=torch.int32, device=dev), cls=torch.empty((nl, cap), dtype=torch.int32, device=dev),
                    box=torch.empty((nl, cap, 4), dtype=torch.float32, device=dev), anchor=torch.empty((nl, cap, 2), dtype=torch.float32, device=dev),
                    count=torch.empty((nl,), dtype=torch.int32, device=dev), cap=cap, shapes=shapes, nt=nt)
        _hip.check(_hip.lib().yp_build_targets(targets.data_ptr(), nt, anchors.data_ptr(), nl, na, geo.data_ptr(), float(self.hyp['anchor_t']), cap,
                                               ents["cell"].data_ptr(), ents["cls"].data_ptr(), ents["box"].data_ptr(), ents["anchor"].data_ptr(),
                                               ents["count"].data_ptr(), _hip.stream_ptr()))
        return ents

    def build_targets(self, p, targets):
        """The reference's return format (loss_functions.py:177-234): per level the classes, boxes (cell offsets + grid wh),
        (image, anchor, gy, gx) index tensors and anchors of the entries.  Slicing the lists to their length reads the counts
        back (one host synchronisation); the loss itself never needs this form."""
        e = self.assign(p, targets)
        tcls, tbox, indices, anch = [], [], [], []
        for l, n in enumerate(e["count"].tolist()):
            _, na, ny, nx = e["shapes"][l][:4]
            c = e["cell"][l, :n].long()
            indices.append((c // (na * ny * nx), (c // (ny * nx)) % na, (c // nx) % ny, c % nx))
            tcls.append(e["cls"][l, :n].long())
            tbox.append(e["box"][l, :n])
            anch.append(e["anchor"][l, :n])
        return tcls, tbox, indices, anch

    # -- the loss ----------------------------------------------------------------------------------------------------------
    def __call__(self, p, targets, prepared=None):
        """`prepared`: the result of assign() computed ahead of time (it depends on the labels and the level shapes only)."""
        from .. import _hip
        if not p[0].is_cuda:
            raise _hip.YpError("ComputeObjectLoss runs on the device only (its CPU statement is test infrastructure, not product)")
        ents = prepared if isinstance(prepared, dict) else self.assign(p, targets)
        return _ObjLossNative.apply(self, ents, *p)


class _ObjLossNative(torch.autograd.Function):
    """Per Detect level: objloss_init, objloss_targets (CIoU through forward-mode dual numbers), objloss_cls, objloss_cells --
    value and gradient together; duplicated (cell, anchor) claims resolve like a sequential index_put (the later entry owns the
    cell).  Pinned to the reference's values and gradients by tests/test_gpu_losses_golden.py."""

    @staticmethod
    @_guarded
    def forward(ctx, owner, ents, *p):
        from .. import _hip
        dev = p[0].device
        sums = torch.zeros(3, dtype=torch.float32, device=dev)
        dps = []
        hyp = owner.hyp
        st = _hip.stream_ptr()
        cap = ents["cap"] if ents["nt"] else 0
        for i, pi in enumerate(p):
            pi = pi.contiguous()
            assert pi.dtype == torch.float32
            no = pi.shape[-1]
            cells = pi.numel() // no
            dp = torch.empty_like(pi)
            iou = torch.empty((max(cap, 1),), dtype=torch.float32, device=dev)
            own = torch.empty((cells,), dtype=torch.int32, device=dev)
            _hip.check(_hip.lib().yp_objloss_level_dev(pi.data_ptr(), cells, no, owner.nc, ents["cell"][i].data_ptr(), ents["box"][i].data_ptr(),
                                                       ents["anchor"][i].data_ptr(), ents["cls"][i].data_ptr(), cap, ents["count"][i:i + 1].data_ptr(),
                                                       float(owner.cp), float(owner.cn), float(hyp['cls_pw']), float(hyp['obj_pw']), float(hyp['box']),
                                                       float(hyp['obj']) * float(owner.balance[i]), float(hyp['cls']), iou.data_ptr(), own.data_ptr(),
                                                       dp.data_ptr(), sums.data_ptr(), st))
            dps.append(dp)
        ctx.dps = dps
        ctx.mark_non_differentiable(sums)
        return sums.sum(dim=0, keepdim=True), sums          # (not a view: the reference scales the loss in place, train.py:238-240)

    @staticmethod
    @_guarded
    def backward(ctx, g, _g_items):
        return (None, None, *torch._foreach_mul(ctx.dps, g.reshape(()).float()))


# ---------------------------------------------------------------------------------------------
# InfoNCE descriptor loss and its geometry helpers (reference utils/utils.py:274-295,333-376 and
# utils/loss_functions.py:339-359,484-597)
# ---------------------------------------------------------------------------------------------
def warp_points(points, homographies, device='cpu'):
    """points [N,2] (x,y) through homographies [3,3] or [B,3,3] -> [N,2] or [B,N,2]."""
    single = homographies.dim() == 2
    H = homographies.unsqueeze(0) if single else homographies
    B = H.shape[0]
    pts = torch.cat((points.float().to(device), torch.ones((points.shape[0], 1), device=device)), dim=1)
    w = (H.reshape(B * 3, 3) @ pts.t()).view(B, 3, -1).transpose(2, 1)
    w = w[:, :, :2] / w[:, :, 2:]
    return w[0] if single else w


def homography_scaling(homography, H, W, device='cpu'):
    """Homography in normalised [-1,1] coordinates -> pixel coordinates of an HxW grid."""
    trans, inv = _const(("hscale", H, W, str(device)), lambda: (lambda t: (t, t.inverse()))(
        torch.tensor([[2. / W, 0., -1.], [0., 2. / H, -1.], [0., 0., 1.]], dtype=torch.float32, device=device)))
    return inv @ homography @ trans


def warp_image_batch(img, mat_homo_inv, device='cpu', mode='bilinear', padding_mode='zeros'):
    """Inverse-warp a batch [B,C,H,W] with normalised inverse homographies [B,3,3] (grid_sample, align_corners=True)."""
    if img.dim() in (2, 3):
        img = img.view(1, 1, img.shape[-2], img.shape[-1])
    if mat_homo_inv.dim() == 2:
        mat_homo_inv = mat_homo_inv.view(1, 3, 3)
    B, _, H, W = img.shape
    def make():
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing='ij')
        return torch.stack((xs, ys), dim=2).to(device).contiguous()
    cells = _const(("warpgrid", H, W, str(device)), make)
    src = warp_points(cells.view(-1, 2), mat_homo_inv, device).view(B, H, W, 2).float()
    return F.grid_sample(img, src, mode=mode, align_corners=True, padding_mode=padding_mode)


def get_coor_cells(Hc, Wc, uv=False, device='cpu'):
    def make():
        ys, xs = torch.meshgrid(torch.arange(Hc), torch.arange(Wc), indexing='ij')
        return torch.stack((xs, ys) if uv else (ys, xs), dim=2).float().view(-1, 2).to(device)
    return _const(("cells", Hc, Wc, bool(uv), str(device)), make)


def normPts(pts, shape):
    return pts / shape * 2 - 1


_SCRATCH = {}


def _scratch(dev, n, E):
    """Persistent per-shape scratch of the native InfoNCE backward (written and consumed inside one backward call, in stream order)."""
    key = (str(dev), n, E)
    if key not in _SCRATCH:
        _SCRATCH[key] = {"w": torch.empty((n, E), dtype=torch.float32, device=dev), "scale": torch.zeros((1,), dtype=torch.float32, device=dev)}
    return _SCRATCH[key]


class _InfoNCENative(torch.autograd.Function):
    """mean_i( logsumexp_j <da_i, db_idx[i][j]>/tau - <da_i, db_i>/tau ) through csrc/losses.hip (yp_infonce_fwd / _bwd): one
    wavefront per anchor gathers the rows it needs; no [n, negs, D] / [n, n] temporaries; backward without atomics.
    Matches the PyTorch formulation to 1e-5 (tests/test_gpu_training.py::test_native_infonce_*) and takes the YOLOPoint-s step
    from 26.1 to 22.7 ms.  YP_NATIVE_INFONCE=0 selects the Gram-matrix formulation."""

    @staticmethod
    @_guarded
    def forward(ctx, da, db, idx, order, offsets, tau):
        from .. import _hip
        da, db = da.contiguous(), db.contiguous()
        n, D = da.shape
        E = idx.shape[1]
        logits = torch.empty((n, E), dtype=torch.float32, device=da.device)     # saved for the backward: owned by this call
        rows = torch.empty((n,), dtype=torch.float32, device=da.device)
        ctx.tau = tau
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # training: the anchor-side gradient (up to the scalar dL/dloss / (tau n)) comes out of the same gather pass as the logits
            lse = torch.empty((n,), dtype=torch.float32, device=da.device)
            dda_u = torch.empty_like(da)
            _hip.check(_hip.lib().yp_infonce_fwd_grad(da.data_ptr(), db.data_ptr(), idx.data_ptr(), n, E, D, 1.0 / tau, logits.data_ptr(), rows.data_ptr(),
                                                      lse.data_ptr(), dda_u.data_ptr(), None, 0, _hip.stream_ptr()))
            ctx.save_for_backward(da, order, offsets, logits, lse, dda_u)
            ctx.fused = True
        else:
            _hip.check(_hip.lib().yp_infonce_fwd(da.data_ptr(), db.data_ptr(), idx.data_ptr(), n, E, D, 1.0 / tau, logits.data_ptr(), rows.data_ptr(),
                                                 _hip.stream_ptr()))
            ctx.fused = False
        return rows.mean()

    @staticmethod
    @_guarded
    def backward(ctx, g):
        from .. import _hip
        if not ctx.fused:
            raise RuntimeError("InfoNCE: backward of a forward that ran without gradient inputs")
        da, order, offsets, logits, lse, dda_u = ctx.saved_tensors
        n, D = da.shape
        E = logits.shape[1]
        scale = (g.float() * (1.0 / (ctx.tau * n))).reshape(1)
        ddb = torch.empty_like(da)
        _hip.check(_hip.lib().yp_infonce_bwd_db(da.data_ptr(), order.data_ptr(), offsets.data_ptr(), logits.data_ptr(), lse.data_ptr(), n, E, D,
                                                scale.data_ptr(), ddb.data_ptr(), None, 0, _hip.stream_ptr()))
        return dda_u * scale, ddb, None, None, None, None


class _InfoNCEPairNative(torch.autograd.Function):
    """_InfoNCENative over ONE [2n, D] tensor of sampled descriptors (rows [0, n): the image's, rows [n, 2n): the warped image's --
    what one yp_points_sample launch over both passes' descriptor maps returns): one gradient tensor for both halves, no zero-padded
    slice gradients (training step, pair mode)."""

    @staticmethod
    @_guarded
    def forward(ctx, dab, idx, order, offsets, tau):
        from .. import _hip
        dab = dab.contiguous()
        n, D = dab.shape[0] // 2, dab.shape[1]
        E = idx.shape[1]
        w = torch.empty((n, E), dtype=torch.float32, device=dab.device)
        rows, lse = torch.empty((n,), dtype=torch.float32, device=dab.device), torch.empty((n,), dtype=torch.float32, device=dab.device)
        grad = torch.empty_like(dab)                    # [dda (unscaled until the backward) | ddb]
        pa, pb = dab.data_ptr(), dab.data_ptr() + 4 * n * D
        _hip.check(_hip.lib().yp_infonce_fwd_grad(pa, pb, idx.data_ptr(), n, E, D, 1.0 / tau, w.data_ptr(), rows.data_ptr(), lse.data_ptr(), grad.data_ptr(),
                                                  None, 0, _hip.stream_ptr()))
        ctx.save_for_backward(dab, order, offsets, w, lse, grad)
        ctx.tau = tau
        return rows.mean()

    @staticmethod
    @_guarded
    def backward(ctx, g):
        from .. import _hip
        dab, order, offsets, w, lse, grad = ctx.saved_tensors
        n, D = dab.shape[0] // 2, dab.shape[1]
        E = w.shape[1]
        scale = (g.float() * (1.0 / (ctx.tau * n))).reshape(1)
        # (out of place: a second backward through this node -- retain_graph, autograd.grad followed by .backward -- must see the saved,
        # unscaled anchor half again and must not overwrite a tensor that was already handed out)
        out = torch.empty_like(grad)
        _hip.check(_hip.lib().yp_infonce_bwd_db(dab.data_ptr(), order.data_ptr(), offsets.data_ptr(), w.data_ptr(), lse.data_ptr(), n, E, D,
                                                scale.data_ptr(), out.data_ptr() + 4 * n * D, None, 0, _hip.stream_ptr()))
        torch.mul(grad[:n], scale, out=out[:n])
        return out, None, None, None, None


class _PointSampleNative(torch.autograd.Function):
    """F.grid_sample(desc, uv, bilinear, align_corners=True) at [B,P] points -> [B,P,D], for a descriptor map whose memory is
    channels-innermost (what the network emits): csrc/losses.hip yp_points_sample_fwd / _bwd, one wavefront per point.  The
    backward of PyTorch's grid_sampler was 1.3 ms of the 19 ms training step (two calls of 640 us); this one is ~40 us per call."""

    @staticmethod
    @_guarded
    def forward(ctx, desc, uv, order=None, offsets=None):
        from .. import _hip
        B, D, H, W = desc.shape
        P = uv.shape[1]
        out = torch.empty((B, P, D), dtype=torch.float32, device=desc.device)
        _hip.check(_hip.lib().yp_points_sample_fwd(desc.data_ptr(), B, H, W, D, uv.data_ptr(), P, out.data_ptr(), None, _hip.stream_ptr()))
        ctx.save_for_backward(uv, order, offsets)
        ctx.dims = (B, D, H, W, P)
        return out

    @staticmethod
    @_guarded
    def backward(ctx, g):
        from .. import _hip
        uv, order, offsets = ctx.saved_tensors
        B, D, H, W, P = ctx.dims
        g = g.contiguous()
        if order is not None:        # cell-sorted (point, tap) list (point_sample_index): one pass, no atomics, no zero-fill, fixed summation order
            gmap = torch.empty((B, H, W, D), dtype=torch.float32, device=g.device)
            _hip.check(_hip.lib().yp_points_sample_bwd_sorted(g.data_ptr(), B, H, W, D, uv.data_ptr(), P, order.data_ptr(), offsets.data_ptr(), None, 0,
                                                              gmap.data_ptr(), None, _hip.stream_ptr()))
        else:
            gmap = torch.zeros((B, H, W, D), dtype=torch.float32, device=g.device)
            _hip.check(_hip.lib().yp_points_sample_bwd(g.data_ptr(), B, H, W, D, uv.data_ptr(), P, gmap.data_ptr(), _hip.stream_ptr()))
        return gmap.permute(0, 3, 1, 2), None, None, None


def _csr(keys, n_buckets, wide):
    """int32 keys [n_items] -> (order: item ids grouped by key, ascending inside a group; offsets [n_buckets + 1]) -- csrc/sampling.hip
    yp_csr_build (a counting sort whose result does not depend on the order of arrival)."""
    from .. import _hip
    dev = keys.device
    order = torch.empty((keys.numel(),), dtype=torch.int32, device=dev)
    offsets = torch.empty((n_buckets + 1,), dtype=torch.int32, device=dev)
    ws = torch.empty((_hip.lib().yp_csr_workspace_ints(keys.numel(), n_buckets),), dtype=torch.int32, device=dev)
    _hip.check(_hip.lib().yp_csr_build(keys.data_ptr(), keys.numel(), n_buckets, 1 if wide else 0, order.data_ptr(), offsets.data_ptr(), ws.data_ptr(),
                                       _hip.stream_ptr()))
    return order, offsets


def point_sample_index(uv, H, W, count_dev=None):
    """uv [B, P, 2] (normalised sample coordinates on an H x W map) -> (order int32 [B*P*4], offsets int32 [B*H*W + 1]): the (point, tap)
    pairs of the bilinear lookup grouped by the cell they touch (ascending inside a cell; taps outside the map / of weight 0 are not
    listed) and the CSR offsets of the cells, for _PointSampleNative's atomic-free backward.  Label-only work (it depends on the sample
    points alone): a training step builds it beside the forward pass."""
    from .. import _hip
    B, P = uv.shape[0], uv.shape[1]
    uv = uv.contiguous()
    keys = torch.empty((B * P * 4,), dtype=torch.int32, device=uv.device)
    # (count_dev: device pointer to the real points-per-image count when uv is a capacity-sized compact array -- _prepare_native(sync=False))
    _hip.check(_hip.lib().yp_points_sample_taps(uv.data_ptr(), B, P, H, W, keys.data_ptr(), count_dev, _hip.stream_ptr()))
    return _csr(keys, B * H * W, wide=False)


def infonce_edges(rnd):
    """rnd [n, negs] (negatives of each match) -> (idx [n, 1+negs] int32 with the match itself in column 0, edge ids grouped by the
    column they point at (ascending inside a group), CSR offsets [n+1]): the backward of the native kernel walks the transposed edge
    list instead of scattering."""
    n = rnd.shape[0]
    idx = torch.cat((torch.arange(n, device=rnd.device).unsqueeze(1), rnd), 1).to(torch.int32).contiguous()
    if idx.is_cuda:
        return (idx,) + _csr(idx.view(-1), n, wide=True)
    skeys, order = torch.sort(idx.flatten(), stable=True)
    offsets = torch.searchsorted(skeys, torch.arange(n + 1, device=rnd.device, dtype=skeys.dtype)).to(torch.int32)
    return idx, order.to(torch.int32), offsets


def _prepare_native(mask_valid_warp, inv_homographies, B, Hc, Wc, samples, negs, pair_index, sync=True):
    """infonce_prepare on the device (csrc/sampling.hip): validity of the cells and their matches, the uniform draw of `pool` cells per
    image, the negatives, the transposed edge list and (pair_index) the cell-sorted tap list -- ~16 native launches, one host
    synchronisation (the common pool size fixes the tensor shapes).  The two Philox keys come from torch's CPU generator:
    torch.manual_seed reproduces the draws."""
    from .. import _hip
    lib, check, sp = _hip.lib(), _hip.check, _hip.stream_ptr
    dev = mask_valid_warp.device
    H, W, Nc = Hc * 8, Wc * 8, Hc * Wc
    mask = mask_valid_warp.float().contiguous()
    inv = inv_homographies.to(dev, torch.float32).contiguous()
    assert mask.numel() == B * H * W and inv.numel() == 9 * B
    seeds = torch.empty((2,), dtype=torch.int64).random_()
    s0, s1 = int(seeds[0]), int(seeds[1])
    valid = torch.empty((B * Nc,), dtype=torch.uint8, device=dev)
    uvb = torch.empty((B * Nc, 2), dtype=torch.float32, device=dev)
    store = torch.empty((2 * B * samples * 2,), dtype=torch.float32, device=dev)
    meta = torch.empty((4,), dtype=torch.int32, device=dev)
    check(lib.yp_nce_cells(mask.data_ptr(), inv.data_ptr(), B, H, W, valid.data_ptr(), uvb.data_ptr(), sp()))
    check(lib.yp_nce_select(valid.data_ptr(), uvb.data_ptr(), B, Hc, Wc, samples, s0, store.data_ptr(), meta.data_ptr(), sp()))
    if not sync:
        # No read-back: every array keeps its CAPACITY (pool = samples) and the counts stay on the device -- meta[0] = points per image,
        # meta[1] = matched rows; the consumers (engine.TrainStep's native loss stage) hand the kernels these pointers.  Returns a dict.
        n_cap, E = B * samples, negs + 1
        idx = torch.empty((n_cap, E), dtype=torch.int32, device=dev)
        check(lib.yp_nce_negatives(n_cap, negs, s1, meta.data_ptr(), idx.data_ptr(), 1, sp()))
        order, offsets = _csr(idx.view(-1), n_cap, wide=True)
        uab = store.view(2 * B, samples, 2)             # (compact [2B][pool][2] at the front of the buffer)
        s_order, s_offsets = point_sample_index(uab, Hc, Wc, count_dev=meta.data_ptr())
        return dict(meta=meta, uab=store, idx=idx, order=order, offsets=offsets, s_order=s_order, s_offsets=s_offsets, n_cap=n_cap, pool_cap=samples, E=E)
    pool = int(meta[0])                                 # (the one host synchronisation)
    if pool <= 0:
        raise _hip.YpError("infonce: an image of the batch has no valid cell")
    n, E = B * pool, negs + 1
    uab = store[:2 * n * 2].view(2 * B, pool, 2)
    idx = torch.empty((n, E), dtype=torch.int32, device=dev)
    check(lib.yp_nce_negatives(n, negs, s1, meta.data_ptr(), idx.data_ptr(), 0, sp()))
    edges = (idx,) + _csr(idx.view(-1), n, wide=True)
    out = (uab[:B], uab[B:], idx[:, 1:], edges)
    if pair_index:
        out = out + ((uab,) + point_sample_index(uab, Hc, Wc),)
    return out


def infonce(descriptors, descriptors_warped, mask_valid_warp, inv_homographies, num_samples_per_image=1500,
            num_masked_non_matches_per_match=120, cell_size=8, device='cpu', tau=0.07, perm_fn=None, randint_fn=None, prepared=None,
            descriptors_pair=None):
    """Cross-image InfoNCE between the descriptors of an image and of its warp (reference utils/loss_functions.py:484-597):
    every valid cell of image A is matched to the cell its inverse homography maps it to in image B;
    `num_masked_non_matches_per_match` random other matches are the negatives;
    loss = -log softmax(<a,b+>/tau | <a,b->/tau)[0], averaged.

    The reference materialises the gathered negatives ([n, negs, D]: 1.5 GB at n = 12000, negs = 120, D = 256) and draws the
    indices with numpy on the host.  Here, on the device, csrc/losses.hip gathers exactly the rows each match needs (forward and
    backward, no [n, negs, D] or [n, n] tensor); elsewhere the negative logits are read out of the Gram matrix da @ db^T.  Unless the draws are injected (`perm_fn` / `randint_fn`: the parity tests replay the reference's draws),
    cells and negatives are drawn on the device with the same distributions (uniform permutation of the valid cells;
    uniform negatives, a negative equal to its own match redrawn from [0, #collisions) exactly as the reference does)."""
    if prepared is None:
        prepared = infonce_prepare(mask_valid_warp, inv_homographies, tuple(descriptors.shape), descriptors.is_cuda, num_samples_per_image,
                                   num_masked_non_matches_per_match, cell_size, device, perm_fn, randint_fn)
    ua, ub, rnd = prepared[:3]
    edges = prepared[3] if len(prepared) > 3 else None

    def sample(desc, idx, inverse=None):
        if (desc.is_cuda and desc.dtype == torch.float32 and desc.shape[1] % 64 == 0 and desc.shape[1] <= 256 and desc.stride(1) == 1
                and desc.permute(0, 2, 3, 1).is_contiguous() and sw("YP_NATIVE_INFONCE") != "0"):
            if inverse is not None:
                return _PointSampleNative.apply(desc, idx.contiguous(), *inverse)
            return _PointSampleNative.apply(desc, idx.contiguous())
        return F.grid_sample(desc, idx.unsqueeze(1), mode='bilinear', align_corners=True).squeeze(2).transpose(1, 2)

    if descriptors_pair is not None and rnd.shape[1] + 1 <= 512 and sw("YP_NATIVE_INFONCE") != "0":
        # both passes' descriptor maps as one [2B, D, Hc, Wc] tensor (image pass first; `descriptors` / `descriptors_warped` are its halves):
        # one sampling launch, one loss call, ONE gradient map for the whole tensor
        # (prepared[4], when present: the pair's sample points and their cell-sorted tap list, built with the sampling)
        pair_inv = prepared[4] if len(prepared) > 4 else None
        dab = sample(descriptors_pair, pair_inv[0], pair_inv[1:]) if pair_inv is not None else sample(descriptors_pair, torch.cat((ua, ub)))
        if edges is None:
            edges = infonce_edges(rnd)
        return _InfoNCEPairNative.apply(dab.flatten(0, 1), *edges, float(tau))
    da = sample(descriptors, ua)                       # [B, pool, D]
    db = sample(descriptors_warped, ub)
    D = da.shape[-1]
    if (da.is_cuda and da.dtype == torch.float32 and D % 64 == 0 and D <= 256 and rnd.shape[1] + 1 <= 512
            and sw("YP_NATIVE_INFONCE") != "0"):
        if edges is None:
            edges = infonce_edges(rnd)
        return _InfoNCENative.apply(da.flatten(0, 1), db.flatten(0, 1), *edges, float(tau))
    pos = (da * db).sum(-1).flatten()
    da, db = da.flatten(0, 1), db.flatten(0, 1)
    neg = (da @ db.t()).gather(1, rnd.long())          # [n, negs] = <da[i], db[rnd[i, j]]>
    logits = torch.cat([pos.unsqueeze(1), neg], dim=1) / tau
    return -F.log_softmax(logits, dim=1)[:, 0].mean()


def infonce_prepare(mask_valid_warp, inv_homographies, desc_shape, on_device, num_samples_per_image=1500,
                    num_masked_non_matches_per_match=120, cell_size=8, device='cpu', perm_fn=None, randint_fn=None, pair_index=False, sync=True):
    """The label-only half of `infonce`: which cells are matched (normalised sample coordinates ua, ub [B, pool, 2]) and
    which matches serve as negatives (rnd [n, negs]).  With `sync` the common pool size is read back (one host synchronisation) and the
    arrays are trimmed to it; `sync=False` (cuda, native sampling) keeps capacity-sized arrays with the counts in device memory (a dict,
    see `_prepare_native`), which is what the training step uses.  An image without a valid cell empties the pool: the
    synchronising form raises; the device-count form cannot (no read-back) and yields a step without a descriptor term (zero rows, zero
    gradient) -- the reference would average over nothing there (NaN)."""
    assert desc_shape[-1] * desc_shape[-2] >= num_samples_per_image, \
        "Number of samples per image must be greater than number of pixels in image"
    with torch.no_grad():
        B, Hc, Wc = desc_shape[0], desc_shape[2], desc_shape[3]
        if (perm_fn is None and randint_fn is None and on_device and mask_valid_warp.is_cuda and cell_size == 8 and Hc * Wc < 36864
                and mask_valid_warp.shape[-2] == 8 * Hc and mask_valid_warp.shape[-1] == 8 * Wc and sw("YP_NATIVE_PREPARE") != "0"):
            return _prepare_native(mask_valid_warp, inv_homographies, B, Hc, Wc, num_samples_per_image, num_masked_non_matches_per_match, pair_index,
                                   sync=sync)
        assert sync, "infonce_prepare(sync=False) exists for the device-side formulation only"
        uv_a = get_coor_cells(Hc, Wc, uv=True).to(device)
        inv_h = inv_homographies.to(device)
        valid = warp_image_batch(mask_valid_warp, inv_h, mode='nearest', device=device)
        valid = (getMasks(valid, device, cell_size) == 1.).flatten(1, -1)                  # [B, Hc*Wc]
        uv_b = warp_points(uv_a, homography_scaling(inv_h, Hc, Wc, device=device), device).round_()   # [B, N, 2]
        if perm_fn is None and on_device:
            # uniform random subset of the valid cells = the cells with the smallest random keys (one host sync for `pool`)
            pool = min(num_samples_per_image, int(valid.sum(1).min()))
            keys = torch.rand(valid.shape, device=valid.device).masked_fill_(~valid, 2.0)
            idx = torch.topk(keys, pool, dim=1, largest=False).indices                     # [B, pool]
            pa = uv_a[idx]
            pb = torch.gather(uv_b, 1, idx.unsqueeze(-1).expand(-1, -1, 2))
        else:
            perm = perm_fn or torch.randperm
            a_list = [uv_a[valid[i]] for i in range(B)]
            b_list = [uv_b[i][valid[i]] for i in range(B)]
            pool = min(num_samples_per_image, min(x.shape[0] for x in a_list))
            pa, pb = [], []
            for i in range(B):
                choice = perm(a_list[i].shape[0])
                pa.append(a_list[i][choice][:pool])
                pb.append(b_list[i][choice][:pool])
            pa, pb = torch.stack(pa).to(device), torch.stack(pb).to(device)
        size = _const(("size", Wc, Hc, str(device)), lambda: torch.tensor([Wc, Hc]).float().to(device))
        ua = normPts(pa, size)
        ub = normPts(pb, size)
        n, negs = ua.shape[0] * ua.shape[1], num_masked_non_matches_per_match
        if randint_fn is None and on_device:
            rnd = torch.randint(0, n, (n, negs), device=ua.device)                     # rnd[i, j]: j-th negative of match i
            same = rnd == torch.arange(n, device=ua.device).unsqueeze(1)               # a negative must not be the match itself
            cand = (torch.rand(rnd.shape, device=ua.device) * same.sum()).long()
            rnd = torch.where(same, cand, rnd)
        else:
            draw = randint_fn or np.random.randint
            shape = (negs, n)
            ordered = np.broadcast_to(np.arange(n), shape)
            rnd = draw(0, n, size=shape)
            same = ordered == rnd
            if nz := np.count_nonzero(same):
                while True:
                    cand = draw(0, nz, nz)
                    if (rnd[same] != cand).any():
                        rnd[same] = cand
                        break
            rnd = torch.from_numpy(np.ascontiguousarray(rnd.T)).to(ua.device)
        edges = infonce_edges(rnd) if (on_device and rnd.is_cuda) else None
        pair_inv = None
        if pair_index and on_device and ua.is_cuda:
            uab = torch.cat((ua, ub)).contiguous()
            pair_inv = (uab,) + point_sample_index(uab, Hc, Wc)
    return (ua, ub, rnd, edges, pair_inv) if pair_inv is not None else (ua, ub, rnd, edges)


descriptor_loss_sparse = infonce      # the name train.py imports it under (train.py:8)


def smooth_BCE(eps=0.1):
    """Label-smoothing BCE targets (positive, negative) -- reference utils/loss_functions.py (YOLOv5's helper), kept for its callers."""
    return 1.0 - 0.5 * eps, 0.5 * eps
