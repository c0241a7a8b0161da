"""Homography-adaptation export (reference src/export_homography.py:23-178): the pseudo-ground-truth generator of the trainer.

For every image the dataset yields N warped views (reference datasets/DataClasses.py:456-489: `image` [1,N,C,H,W], `valid_mask`
[N,1,H,W], `inv_homographies` [1,N,3,3], optional `pad` / `dims`).  The views go through the network as ONE batch (a native plan of
batch N, 16-bit), the keypoint head is decoded on the device (flattenDetection), the N heat maps are carried back to the base
frame and averaged by one kernel (csrc/postproc.hip `yp_homo_combine`: no 2 x N warped full-resolution temporaries), and the
aggregated map is thresholded / grid-NMS'd / border-filtered on the device.  The only host transfer per image is the point list,
written as `<name>.npz {'pts': [n,3] (x, y, prob)}` -- the format DataClasses.py:179-189 reads back.

    exp = HomographyExporter(model, device, dict(nms=4, top_k=600, detection_threshold=0.015))
    pts = exp.export_sample(sample)                # numpy [n,3], conf descending
    homographic_export(config, samples, model, output_dir, export_task="train")   # the reference's loop over a sample iterable
"""
from pathlib import Path

import numpy as np
import torch

from . import _hip
from .utils import utils as U

HA_DEFAULT = dict(nms=4, top_k=600, detection_threshold=0.015)           # reference configs/*_export.yaml: homography_adaptation


@_hip.guarded
def combine_heatmaps(heat, mask, inv_homographies, want_cover=False):
    """heat, mask: cuda fp32 [N,H,W] (or [N,1,H,W]); inv_homographies [N,3,3] (normalised coordinates)
    -> sum_v warp(heat_v * mask_v) / sum_v warp(mask_v), cuda [H,W]  (export_homography.py:94-96,143-145)."""
    heat = U.as_cuda_f32(heat, what="heat").reshape(-1, heat.shape[-2], heat.shape[-1]).contiguous()
    mask = U.as_cuda_f32(mask, what="mask").reshape(-1, mask.shape[-2], mask.shape[-1]).contiguous()
    inv_h = U.as_cuda_f32(inv_homographies, what="inv_homographies").reshape(-1, 9).contiguous()
    N, H, W = heat.shape
    if mask.shape != heat.shape or inv_h.shape[0] != N:
        raise _hip.YpError(f"combine_heatmaps: {tuple(heat.shape)} heat maps, {tuple(mask.shape)} masks, {inv_h.shape[0]} homographies")
    out = torch.empty((H, W), dtype=torch.float32, device=heat.device)
    cover = torch.empty((H, W), dtype=torch.float32, device=heat.device) if want_cover else None
    _hip.check(_hip.lib().yp_homo_combine(heat.data_ptr(), mask.data_ptr(), inv_h.data_ptr(), N, H, W, out.data_ptr(),
                                          cover.data_ptr() if want_cover else None, _hip.stream_ptr()))
    return (out, cover) if want_cover else out


class HomographyExporter:
    def __init__(self, model, device, ha_config=None, normalize_points=False):
        self.model, self.device = model.eval(), torch.device(device)
        cfg = dict(HA_DEFAULT, **(ha_config or {}))
        self.nms_dist, self.top_k, self.conf_thresh = int(cfg["nms"]), cfg.get("top_k"), float(cfg["detection_threshold"])
        self.normalize_points = normalize_points

    @torch.no_grad()
    @_hip.guarded
    def aggregate(self, views, valid_mask, inv_homographies, pad=None):
        """views [N,C,H,W]; valid_mask [N,1,H,W] or [N,H,W]; inv_homographies [N,3,3] -> aggregated heat map, cuda [H',W']."""
        views = views.to(self.device)
        semi = self.model(views)["semi"]
        heat = U.flattenDetection(semi)                          # [N,1,H,W] on the device
        out = combine_heatmaps(heat, valid_mask, inv_homographies)
        if pad is not None:                                      # export_homography.py:100-106, index arithmetic kept as written
            pad = [int(v) for v in pad]
            height, width = out.shape
            if pad[1]:
                out = out[pad[0]:width - pad[1], :]
            if pad[3]:
                out = out[:, pad[2]:height - pad[3]]
        return out

    @torch.no_grad()
    @_hip.guarded
    def export_sample(self, sample):
        """One dataset sample (reference layout, batch dimension of 1 in front) -> numpy [n,3] (x, y, prob)."""
        img = sample["image"]
        img = img.squeeze(0) if img.dim() == 5 else img
        inv_h = sample["inv_homographies"]
        inv_h = inv_h[0] if inv_h.dim() == 4 else inv_h
        mask = sample["valid_mask"]
        mask = mask.squeeze(0) if mask.dim() == 5 else mask      # the reference's transpose(0,1) of [1,N,H,W] gives [N,1,H,W]
        if mask.dim() == 4 and mask.shape[0] == 1 and mask.shape[1] == img.shape[0]:
            mask = mask.transpose(0, 1)
        agg = self.aggregate(img, mask, inv_h, sample.get("pad"))
        pts = U.getPtsFromHeatmap(agg.contiguous(), self.conf_thresh, self.nms_dist).transpose()
        if self.top_k and pts.shape[0] > self.top_k:
            pts = pts[:self.top_k, :]
        if self.normalize_points:
            (H, W) = sample["dims"][1]
            pts[:, 0] /= float(W)
            pts[:, 1] /= float(H)
        return pts


def homographic_export(config, samples, model, output_dir, export_task="train", device="cuda:0"):
    """The reference's export loop (export_homography.py:72-178) over an iterable of samples; returns the written paths.
    With torch.distributed initialised, rank r handles samples r, r + world, ... (the images are independent: no collective)."""
    ha = config["data"]["homography_adaptation"]
    exp = HomographyExporter(model.to(device), device, dict(nms=ha["nms"], top_k=ha.get("top_k"), detection_threshold=ha["detection_threshold"]),
                             normalize_points=bool(config.get("normalize_points")))
    out_dir = Path(output_dir) / export_task
    out_dir.mkdir(parents=True, exist_ok=True)
    rank, world = 0, 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    written = []
    for i, sample in enumerate(samples):
        if i % world != rank:
            continue
        name = sample["name"][0] if isinstance(sample["name"], (list, tuple)) else sample["name"]
        pts = exp.export_sample(sample)
        path = out_dir / f"{name}.npz"
        np.savez_compressed(path, pts=pts)
        written.append(path)
    return written
