"""Pairwise box overlap (reference: src/utils/metrics_yolo.py:243-265).  The CIoU of the object loss is computed inside
csrc/losses.hip (objloss_targets_kernel); its PyTorch statement lives with the parity tests."""
import torch


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of xyxy boxes [N,4] x [M,4] -> [N,M] (reference :243-265)."""
    (a1, a2), (b1, b2) = box1[:, None].chunk(2, 2), box2.chunk(2, 1)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(1) - inter + eps)
