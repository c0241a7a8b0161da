"""Pairwise box overlap (reference: src/utils/metrics_yolo.py:243-265).  The CIoU of the object loss is computed inside
csrc/losses.hip (objloss_targets_kernel); its PyTorch statement lives with the parity tests."""
import torch


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of xyxy boxes [N,4] x [M,4] -> [N,M] (reference :243-265)."""
    (a1, a2), (b1, b2) = box1[:, None].chunk(2, 2), box2.chunk(2, 1)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(1) - inter + eps)


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU (or its GIoU / DIoU / CIoU penalised forms) of row-paired boxes [N,4] x [N,4] -> [N,1]: the public helper of reference
    utils/metrics_yolo.py:202-240, kept for its callers (evaluation scripts, custom losses).  Plain torch, any device.  The training step
    does not come through here: the object loss evaluates CIoU and its gradient inside csrc/losses.hip.

    Worked in centre / extent form: a box is (c, e) with c its centre and e its (w, h); overlap along an axis is
    max(0, min(hi1, hi2) - max(lo1, lo2)) with lo/hi = c -/+ e/2, the enclosing box spans max(hi) - min(lo)."""
    import math

    def corners_extent(b):
        """(lo, hi, extent, centre); corner boxes keep their corners as given (no clamping of a degenerate height, as the reference)."""
        if xywh:
            c, e = b[..., :2], b[..., 2:4]
            return c - e * 0.5, c + e * 0.5, e, c
        lo, hi = b[..., :2], b[..., 2:4]
        return lo, hi, hi - lo, (lo + hi) * 0.5
    (lo1, hi1, e1, c1), (lo2, hi2, e2, c2) = corners_extent(box1), corners_extent(box2)
    inter = (torch.minimum(hi1, hi2) - torch.maximum(lo1, lo2)).clamp(0).prod(-1, keepdim=True)
    union = e1.prod(-1, keepdim=True) + e2.prod(-1, keepdim=True) - inter + eps
    iou = inter / union
    if not (GIoU or DIoU or CIoU):
        return iou
    span = torch.maximum(hi1, hi2) - torch.minimum(lo1, lo2)            # enclosing box extent
    if GIoU and not (DIoU or CIoU):
        hull = span.prod(-1, keepdim=True) + eps
        return iou - (hull - union) / hull
    centre_term = ((c2 - c1) ** 2).sum(-1, keepdim=True) / ((span ** 2).sum(-1, keepdim=True) + eps)
    if not CIoU:
        return iou - centre_term
    aspect = (4.0 / math.pi ** 2) * (torch.atan(e2[..., :1] / (e2[..., 1:] + eps)) - torch.atan(e1[..., :1] / (e1[..., 1:] + eps))) ** 2
    with torch.no_grad():
        weight = aspect / (aspect - iou + (1.0 + eps))
    return iou - (centre_term + aspect * weight)
