"""Box-overlap helpers used by the object loss (reference: src/utils/metrics_yolo.py:202-265)."""
import math

import torch


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU / GIoU / DIoU / CIoU of box1 [n,4] against box2 [n,4] -> [n,1] (reference :202-240)."""
    if xywh:
        (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, 1), box2.chunk(4, 1)
        a_x1, a_x2, a_y1, a_y2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
        b_x1, b_x2, b_y1, b_y2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    else:
        a_x1, a_y1, a_x2, a_y2 = box1.chunk(4, 1)
        b_x1, b_y1, b_x2, b_y2 = box2.chunk(4, 1)
        w1, h1, w2, h2 = a_x2 - a_x1, a_y2 - a_y1, b_x2 - b_x1, b_y2 - b_y1
    inter = (torch.min(a_x2, b_x2) - torch.max(a_x1, b_x1)).clamp(0) * (torch.min(a_y2, b_y2) - torch.max(a_y1, b_y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not (CIoU or DIoU or GIoU):
        return iou
    cw = torch.max(a_x2, b_x2) - torch.min(a_x1, b_x1)          # smallest enclosing box
    ch = torch.max(a_y2, b_y2) - torch.min(a_y1, b_y1)
    if CIoU or DIoU:
        c2 = cw ** 2 + ch ** 2 + eps
        rho2 = ((b_x1 + b_x2 - a_x1 - a_x2) ** 2 + (b_y1 + b_y2 - a_y1 - a_y2) ** 2) / 4
        if CIoU:
            v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / (h2 + eps)) - torch.atan(w1 / (h1 + eps)), 2)
            with torch.no_grad():
                alpha = v / (v - iou + (1 + eps))
            return iou - (rho2 / c2 + v * alpha)
        return iou - rho2 / c2
    c_area = cw * ch + eps
    return iou - (c_area - union) / c_area


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of xyxy boxes [N,4] x [M,4] -> [N,M] (reference :243-265)."""
    (a1, a2), (b1, b2) = box1[:, None].chunk(2, 2), box2.chunk(2, 1)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(1) - inter + eps)
