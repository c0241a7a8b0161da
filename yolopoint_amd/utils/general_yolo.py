"""Box post-processing entry points (reference: src/utils/general_yolo.py:124-235
non_max_suppression, :623-630 xywh2xyxy, :534-536 make_divisible, :46-53 LOGGER)."""
import ctypes as C
import logging
import math
import os

import numpy as np
import torch

from .. import _hip
from ._ws import workspace, as_cuda_f32


def set_logging(name=None, verbose=True):
    rank = int(os.getenv('RANK', -1))
    logging.basicConfig(format="%(message)s", level=logging.INFO if (verbose and rank in (-1, 0)) else logging.WARNING)
    return logging.getLogger(name)


LOGGER = set_logging('yolov5')


def make_divisible(x, divisor):
    """Smallest multiple of `divisor` that is >= x."""
    return math.ceil(x / divisor) * divisor


def xywh2xyxy(x):
    """[cx, cy, w, h] rows -> [x1, y1, x2, y2] rows (torch tensor or numpy array, any device)."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


@_hip.guarded
def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nm=0):
    """Batched NMS on decoded predictions [B, N, 5+nc]; returns a list of B tensors [n, 6]
    (x1, y1, x2, y2, conf, cls), sorted by confidence, exactly as the reference.

    The whole chain (objectness filter, conf = obj*cls, multi-label expansion or best class,
    sort, class offset, greedy IoU suppression, max_det) runs in csrc/postproc.hip; the greedy
    step is torchvision.ops.nms's definition."""
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if nm != 0:
        raise _hip.YpError("non_max_suppression: mask outputs (nm > 0) do not exist on the YOLOPoint path (Detect has no mask head)")
    pred = as_cuda_f32(prediction, what="prediction")
    B, N, no = pred.shape
    nc = no - 5
    if labels and any(len(lb) for lb in labels):
        # a-priori labels (autolabelling, general_yolo.py:171-178): rows [class, x, y, w, h] become predictions with objectness 1 and a
        # one-hot class, appended behind the image's own rows (rows padded with objectness 0 never become candidates)
        extra = max(len(lb) for lb in labels)
        grown = torch.zeros((B, N + extra, no), dtype=torch.float32, device=pred.device)
        grown[:, :N] = pred
        for b, lb in enumerate(labels):
            if len(lb):
                lb = torch.as_tensor(lb, dtype=torch.float32, device=pred.device)
                rows = torch.arange(N, N + lb.shape[0], device=pred.device)
                grown[b, rows, 0:4] = lb[:, 1:5]
                grown[b, rows, 4] = 1.0
                grown[b, rows, 5 + lb[:, 0].long()] = 1.0
        pred, N = grown, N + extra
    mask = None
    if classes is not None:
        # keep only the listed classes (general_yolo.py:199-200): a bit mask tested where the candidates are collected
        words = [0] * ((nc + 31) // 32)
        for c in classes:
            c = int(c)
            if 0 <= c < nc:
                words[c >> 5] |= 1 << (c & 31)
        mask = torch.tensor([w - (1 << 32) if w >= (1 << 31) else w for w in words], dtype=torch.int32, device=pred.device) if words else None
    max_wh, max_nms = 7680.0, 30000     # reference constants (general_yolo.py:154-155)
    l = _hip.lib()
    out = torch.empty((B, max_det, 6), dtype=torch.float32, device=pred.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=pred.device)
    ml = int(bool(multi_label) and nc > 1)
    nbytes = l.yp_box_nms_workspace_bytes(B, N, nc, ml, max_nms)
    ws = workspace(pred.device, nbytes, "box_nms")
    _hip.check(l.yp_box_nms_classes(pred.data_ptr(), B, N, nc, float(conf_thres), float(iou_thres), ml, int(bool(agnostic)),
                                    int(max_det), max_nms, max_wh, mask.data_ptr() if mask is not None else None, out.data_ptr(), cnt.data_ptr(),
                                    ws.data_ptr(), ws.numel(), _hip.stream_ptr()))
    counts = cnt.cpu().tolist()
    if any(c < 0 for c in counts):
        raise _hip.YpError("non_max_suppression: candidate list overflowed the workspace (more than 2^21 candidates per image)")
    return [out[i, :counts[i]].clone() for i in range(B)]
