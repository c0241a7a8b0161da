"""Keypoint decode / NMS entry points and small helpers with the reference's signatures
(reference: src/utils/utils.py:55-57 load_model, :94-101 getPtsFromSemi, :103-116 getMasks,
:118-182 nms_fast, :184-209 labels2Dto3D, :232-262 flattenDetection, :465-485 getPtsFromHeatmap).

The numeric work runs in csrc/postproc.hip through the C ABI; this file only adapts argument
and return conventions (numpy float64 [3, N] point lists, torch heat maps, empty-input shapes).
"""
import ctypes as C
from importlib import import_module

import numpy as np
import torch

from .. import _hip
from ._ws import workspace, as_cuda_f32

BORDER_REMOVE = 4     # reference: utils/utils.py:466


def load_model(meta_model=True, **kwargs):
    """getattr(models, name)(**kwargs)  (reference: utils/utils.py:55-57)."""
    model_name = 'Model' if meta_model else kwargs.pop('model_name')
    return getattr(import_module('yolopoint_amd.models'), model_name)(**kwargs)


def toNumpy(tensor):
    return tensor.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------
@_hip.guarded
def _kp_decode(semi4, mode):
    """semi4: fp32 cuda tensor [B,65,Hc,Wc] with arbitrary strides -> heat [B, 8Hc, 8Wc]."""
    B, ch, Hc, Wc = semi4.shape
    if ch != 65:
        raise _hip.YpError(f"flattenDetection: expected 65 channels (cell_size 8), got {ch}")
    heat = torch.empty((B, Hc * 8, Wc * 8), dtype=torch.float32, device=semi4.device)
    sb, sc, sy, sx = semi4.stride()
    _hip.check(_hip.lib().yp_kp_decode(semi4.data_ptr(), B, Hc, Wc, sb, sc, sy, sx, mode, heat.data_ptr(),
                                       _hip.stream_ptr()))
    return heat


def flattenDetection(semi, cell_size=8):
    """softmax over the 65 cell channels, drop the dustbin, depth-to-space(8).
    semi [65,Hc,Wc] -> [1,H,W];  semi [B,65,Hc,Wc] -> [B,1,H,W]."""
    if cell_size != 8:
        raise _hip.YpError("flattenDetection: the decode kernel is specialised for cell_size=8")
    batch = semi.dim() == 4
    dev = semi.device
    s = semi if semi.is_cuda and semi.dtype == torch.float32 else as_cuda_f32(semi.detach(), what="semi")
    heat = _kp_decode(s if batch else s.unsqueeze(0), 0)
    heat = heat.unsqueeze(1) if batch else heat
    return heat if dev.type == "cuda" else heat.to(dev)


def flattenDetection_demo(semi):
    """The demo's variant exp(x)/(sum+1e-5) without max subtraction (reference: demo.py:140-150).
    semi [65,Hc,Wc] or [1,65,Hc,Wc] -> cuda heat map [H,W]."""
    s = as_cuda_f32(semi, what="semi")
    if s.dim() == 3:
        s = s.unsqueeze(0)
    return _kp_decode(s, 1)[0]


# ---------------------------------------------------------------------------------------------
KP_NMS_ROUNDS = 16


@_hip.guarded
def _kp_nms_device(heat3, conf_thresh, radius, border):
    """heat3: fp32 cuda [B,H,W] -> list of B float32 cuda tensors [n,3] (x,y,conf), conf descending."""
    heat3 = heat3.contiguous()
    B, H, W = heat3.shape
    l = _hip.lib()
    step = radius + 1
    max_out = max(1, -(-H // step) * -(-W // step))   # kept points are pairwise > radius apart
    out = torch.empty((B, max_out, 3), dtype=torch.float32, device=heat3.device)
    cnt = torch.empty((B + 1,), dtype=torch.int32, device=heat3.device)     # [counts..., candidates left undecided]
    nbytes = l.yp_kp_nms_workspace_bytes(B, H, W)
    ws = workspace(heat3.device, nbytes, "kp_nms")
    args = (heat3.data_ptr(), B, H, W, float(conf_thresh), int(radius), int(border), out.data_ptr(), cnt.data_ptr(), max_out, ws.data_ptr(), ws.numel())
    # a fixed number of fix-point rounds without a host synchronisation; whether they sufficed is read back with the counts
    _hip.check(l.yp_kp_nms_async(*args, KP_NMS_ROUNDS, cnt[B:].data_ptr(), _hip.stream_ptr()))
    counts = cnt.cpu().tolist()
    if counts[B]:                                                          # not converged: the synchronising variant iterates to the fix-point
        _hip.check(l.yp_kp_nms(*args, _hip.stream_ptr()))
        counts = cnt.cpu().tolist()
    return [out[b, :counts[b]] for b in range(B)]


def getPtsFromHeatmap(heatmap, conf_thresh, nms_dist):
    """heat map [H,W] (numpy or torch) -> float64 numpy [3,N] rows (x, y, conf), conf descending:
    threshold (>=) -> greedy grid NMS radius nms_dist -> drop points within 4 px of the border."""
    h = as_cuda_f32(heatmap, what="heatmap")
    if h.dim() != 2:
        raise _hip.YpError(f"getPtsFromHeatmap: expected [H,W], got {tuple(h.shape)}")
    pts = _kp_nms_device(h.unsqueeze(0), conf_thresh, nms_dist, BORDER_REMOVE)[0]
    if pts.shape[0] == 0:
        return np.zeros((3, 0))
    return pts.cpu().numpy().astype(np.float64).T.copy()


def getPtsFromSemi(semi, conf_thresh=0.015, nms_dist=4):
    """semi [65,Hc,Wc] (raw head output, no batch) -> [3,N] points; decode and NMS stay on the GPU."""
    heat = flattenDetection(as_cuda_f32(semi.detach(), what="semi"))
    return getPtsFromHeatmap(heat.squeeze(), conf_thresh, nms_dist)


def nms_fast(in_corners, H, W, dist_thresh):
    """Greedy grid NMS of an explicit corner list 3xN (x, y, conf) -> (3xN' survivors sorted by conf desc,
    their indices into in_corners).  Coordinates are rounded to the grid first; when several corners
    round to the same pixel the highest-confidence one competes, and — like the reference's index
    grid, which the last-written duplicate wins — the lowest-confidence duplicate is what is reported."""
    in_corners = np.asarray(in_corners)
    n = in_corners.shape[1]
    if n == 0:
        return np.zeros((3, 0)).astype(int), np.zeros(0).astype(int)
    order = np.argsort(-in_corners[2, :], kind="stable")
    corners = in_corners[:, order]
    rc = corners[:2, :].round().astype(int)
    if n == 1:
        return np.vstack((rc, in_corners[2])).reshape(3, 1), np.zeros((1)).astype(int)
    grid = np.full((H, W), -np.inf, dtype=np.float32)
    last = np.zeros((H, W), dtype=np.int64)
    # duplicates: priority = first (highest conf) visit; reported index = last (lowest conf) write
    np.maximum.at(grid, (rc[1], rc[0]), corners[2].astype(np.float32))
    last[rc[1], rc[0]] = np.arange(n)
    # the grid carries fp32 scores: identical fp32 values would tie, so rank them instead
    rank = np.empty(n, dtype=np.float32)
    rank[:] = n - np.arange(n)
    grid2 = np.full((H, W), -1.0, dtype=np.float32)
    np.maximum.at(grid2, (rc[1], rc[0]), rank)
    kept = _kp_nms_device(torch.from_numpy(grid2).cuda().unsqueeze(0), 0.5, dist_thresh, 0)[0].cpu().numpy()
    ky, kx = kept[:, 1].astype(int), kept[:, 0].astype(int)
    inds_keep = last[ky, kx]
    out = corners[:, inds_keep]
    inds2 = np.argsort(-out[-1, :], kind="stable")
    return out[:, inds2], order[inds_keep[inds2]]


# ---------------------------------------------------------------------------------------------
def labels2Dto3D(labels, cell_size=8, add_dustbin=True):
    """[B,1,H,W] keypoint map -> [B,65,Hc,Wc] cell labels (PixelUnshuffle + dustbin + normalise);
    the inverse arrangement of flattenDetection.  Plain tensor reshuffling on the caller's device."""
    B, _, H, W = labels.shape
    Hc, Wc = H // cell_size, W // cell_size
    cells = labels.reshape(B, 1, Hc, cell_size, Wc, cell_size).permute(0, 1, 3, 5, 2, 4).reshape(
        B, cell_size * cell_size, Hc, Wc)
    if add_dustbin:
        dust = 1 - cells.sum(dim=1)
        dust = torch.where(dust < 1., torch.zeros_like(dust), dust)      # (no boolean-mask assignment: that synchronises)
        cells = torch.cat((cells, dust.view(B, 1, Hc, Wc)), dim=1)
        cells = cells.div(cells.sum(dim=1).unsqueeze(1))
    return cells


def getMasks(mask_2D, device, cell_size=8):
    """[B,1,H,W] validity mask -> [B,Hc,Wc]: a cell is valid when all of its 64 pixels are."""
    return torch.prod(labels2Dto3D(mask_2D.to(device), cell_size=cell_size, add_dustbin=False).float(), 1)
