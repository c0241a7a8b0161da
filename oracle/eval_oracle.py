"""numpy restatement of the reference's evaluation metrics — the definitions of "kp repeatability" and "det mAP"
in BASELINE.json's metric.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  compute_repeatability   evaluations/detector_evaluation.py:27-162   (k = 300 best points, 3 px)
  process_batch           evaluations/yolo_evaluation.py:72-94        (TP matrix over IoU 0.5:0.95)
  ap_per_class/compute_ap utils/metrics_yolo.py:29-121                (101-point interpolated AP)
Pinned by tests/golden/eval.npz (tests/golden/make_golden.py imports the reference to produce it).
"""
import numpy as np


def homography_scaling(Hn, H, W):
    """normalised [-1,1] homography -> pixel homography of an H x W image (utils/utils.py:292-295), float32 like the reference."""
    t = np.array([[2. / W, 0., -1.], [0., 2. / H, -1.], [0., 0., 1.]], dtype=np.float32)
    return (np.linalg.inv(t) @ np.asarray(Hn, dtype=np.float32) @ t).astype(np.float32)


def warp_keypoints(kp, Hom, shape, scale=True):
    """kp [N,(x,y)] through Hom (detector_evaluation.py:27-40)."""
    if scale:
        Hom = homography_scaling(Hom, *shape[:2])
    pts = np.concatenate([kp, np.ones((kp.shape[0], 1))], axis=1)
    w = np.dot(pts, np.transpose(np.asarray(Hom)))
    return w[:, :2] / w[:, 2:]


def compute_repeatability(data, keep_k_points=300, distance_thresh=3, scale=True):
    """data: image [3,H,W] or [H,W,3], homography / inv_homography (normalised), prob / warped_prob [N,3] (x,y,conf).
    Returns (repeatability, localisation error).  Quirk kept from the reference: the warped detections are filtered
    with margin = int(scale) because `scale` is passed in the margin position (detector_evaluation.py:112)."""
    shape = data['image'].shape
    if shape[0] == 3:
        shape = (*shape[1:], 3)
    H, H_inv = data['homography'], data['inv_homography']
    kp = data['prob'].copy()
    wkp = data['warped_prob'].copy()

    def inside(points, margin):
        return (points[:, 0] >= margin) & (points[:, 0] < shape[1] - margin) & (points[:, 1] >= margin) & (points[:, 1] < shape[0] - margin)

    def k_best(points, k):
        if points.shape[1] > 2:
            points = points[points[:, 2].argsort(), :2]
            points = points[-min(k, points.shape[0]):, :]
        return points
    wkp = wkp[inside(warp_keypoints(wkp[:, [0, 1]], H, shape[:2], scale), int(scale))]
    kp[:, :2] = warp_keypoints(kp[:, :2], H_inv, shape)
    kp = kp[inside(kp, 2)]
    wkp, kp = k_best(wkp, keep_k_points), k_best(kp, keep_k_points)
    N1, N2 = kp.shape[0], wkp.shape[0]
    norm = np.linalg.norm(kp[:, None] - wkp[None], ord=None, axis=2)
    c1 = c2 = 0
    e1 = e2 = None
    if N2 != 0:
        m1 = np.min(norm, axis=1)
        c1, e1 = np.sum(m1 <= distance_thresh), m1[m1 <= distance_thresh]
    if N1 != 0:
        m2 = np.min(norm, axis=0)
        c2, e2 = np.sum(m2 <= distance_thresh), m2[m2 <= distance_thresh]
    rep = (c1 + c2) / (N1 + N2) if N1 + N2 > 0 else []
    err = -1
    if c1 + c2 > 0:
        err = 0
        if e1 is not None:
            err += e1.sum() / (c1 + c2)
        if e2 is not None:
            err += e2.sum() / (c1 + c2)
    else:
        rep = 0
    return rep, err


def box_iou(a, b, eps=1e-7):
    """pairwise IoU of xyxy boxes [N,4] x [M,4] (utils/metrics_yolo.py:243-265), fp32."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = np.clip(rb - lt, 0, None).prod(2)
    return inter / ((a[:, 2:] - a[:, :2]).prod(1)[:, None] + (b[:, 2:] - b[:, :2]).prod(1)[None] - inter + np.float32(eps))


def process_batch(detections, labels, iouv):
    """detections [N,6] (xyxy, conf, cls), labels [M,5] (cls, xyxy) -> bool [N, len(iouv)]: a detection is correct at an
    IoU level if it is the best (highest-IoU) match of a label of its class, each label matched at most once."""
    correct = np.zeros((detections.shape[0], iouv.shape[0]), dtype=bool)
    iou = box_iou(labels[:, 1:], detections[:, :4])
    same = labels[:, 0:1] == detections[:, 5]
    for i, thr in enumerate(iouv):
        li, di = np.where((iou >= thr) & same)
        if li.shape[0]:
            m = np.stack((li, di, iou[li, di]), 1).astype(np.float64)
            if li.shape[0] > 1:
                m = m[m[:, 2].argsort()[::-1]]
                m = m[np.unique(m[:, 1], return_index=True)[1]]
                m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1].astype(int), i] = True
    return correct


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    return float(np.sum((y[1:] + y[:-1]) * 0.5 * np.diff(x))), mpre, mrec        # trapezoid rule == np.trapz


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """-> (ap [nc, n_iou], unique classes).  AP per class and IoU level from the confidence-sorted TP matrix."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, nt = np.unique(target_cls, return_counts=True)
    ap = np.zeros((classes.shape[0], tp.shape[1]))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        if sel.sum() == 0 or nt[ci] == 0:
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(0), tp[sel].cumsum(0)
        recall = tpc / (nt[ci] + eps)
        precision = tpc / (tpc + fpc)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    return ap, classes.astype(int)
