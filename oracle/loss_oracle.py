"""CPU restatement (PyTorch autograd) of the YOLO object loss of the reference training step.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): the product computes this loss in csrc/losses.hip (yp_build_targets, yp_objloss_level_dev) and has
no CPU path; this file is what those kernels are checked against, and it is itself pinned to the reference's values and
gradients (tests/golden/losses.npz: obj.*, obj2.*, iou.ciou -- captured by importing the reference, tests/golden/make_golden.py).

Restates /root/reference/src/utils/loss_functions.py:90-234 (ComputeObjectLoss.__init__/__call__/build_targets, reference configs
only: no focal loss, no autobalance, gr = 1) and src/utils/metrics_yolo.py:202-240 (bbox_iou, CIoU branch).
"""
import math

import torch
import torch.nn.functional as F

NEIGHBOUR_OFFSETS = ((0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (-0.5, 0.0), (0.0, -0.5))     # loss_functions.py:189-195 (x g = 0.5)


def ciou(pred_xywh, target_xywh, eps=1e-7):
    """Complete-IoU of centre-format boxes, row by row (metrics_yolo.py:202-240 with xywh=True, CIoU=True) -> [n]."""
    px, py, pw, ph = pred_xywh.unbind(1)
    tx, ty, tw, th = target_xywh.unbind(1)
    pl, pr, pt, pb = px - pw / 2, px + pw / 2, py - ph / 2, py + ph / 2
    tl, tr, tt, tb = tx - tw / 2, tx + tw / 2, ty - th / 2, ty + th / 2
    overlap = (torch.min(pr, tr) - torch.max(pl, tl)).clamp(0) * (torch.min(pb, tb) - torch.max(pt, tt)).clamp(0)
    union = pw * ph + tw * th - overlap + eps
    iou = overlap / union
    hull_w, hull_h = torch.max(pr, tr) - torch.min(pl, tl), torch.max(pb, tb) - torch.min(pt, tt)
    diag2 = hull_w ** 2 + hull_h ** 2 + eps
    centre2 = ((tl + tr - pl - pr) ** 2 + (tt + tb - pt - pb) ** 2) / 4
    aspect = (4 / math.pi ** 2) * (torch.atan(tw / (th + eps)) - torch.atan(pw / (ph + eps))) ** 2
    with torch.no_grad():
        alpha = aspect / (aspect - iou + (1 + eps))
    return iou - (centre2 / diag2 + aspect * alpha)


def assign_targets(labels, anchors, level_shapes, anchor_t):
    """labels [nt,6] (image, class, xc, yc, w, h normalised), anchors [nl,na,2] in grid units, level_shapes [(ny, nx)] ->
    per level a dict of entry arrays in the reference's order (loss_functions.py:177-234): offset-major, then anchor, then label.
    Plain loops: this is the sequential statement the scan kernel is compared with."""
    out = []
    nt, na = labels.shape[0], anchors.shape[1]
    for l, (ny, nx) in enumerate(level_shapes):
        rows = []
        for o, (ox, oy) in enumerate(NEIGHBOUR_OFFSETS):
            for a in range(na):
                for k in range(nt):
                    img, c, x, y, w, h = (float(v) for v in labels[k])
                    gx, gy, gw, gh = torch.tensor([x * nx, y * ny, w * nx, h * ny], dtype=torch.float32).tolist()
                    aw, ah = float(anchors[l, a, 0]), float(anchors[l, a, 1])
                    rw, rh = torch.tensor(gw) / aw, torch.tensor(gh) / ah
                    if not float(torch.max(torch.max(rw, 1 / rw), torch.max(rh, 1 / rh))) < anchor_t:
                        continue
                    fx, fy = torch.tensor(gx, dtype=torch.float32), torch.tensor(gy, dtype=torch.float32)
                    ix, iy = nx - fx, ny - fy
                    ok = (True, bool((fx % 1 < 0.5) & (fx > 1)), bool((fy % 1 < 0.5) & (fy > 1)), bool((ix % 1 < 0.5) & (ix > 1)),
                          bool((iy % 1 < 0.5) & (iy > 1)))[o]
                    if not ok:
                        continue
                    gi = min(max(int((fx - ox).item()), 0), nx - 1)
                    gj = min(max(int((fy - oy).item()), 0), ny - 1)
                    rows.append((int(img), a, gj, gi, int(c), float(fx) - gi, float(fy) - gj, gw, gh, aw, ah))
        t = torch.tensor(rows, dtype=torch.float64).view(-1, 11)
        out.append(dict(b=t[:, 0].long(), a=t[:, 1].long(), gj=t[:, 2].long(), gi=t[:, 3].long(), cls=t[:, 4].long(),
                        box=t[:, 5:9].float(), anchor=t[:, 9:11].float()))
    return out


def object_loss(preds, labels, anchors, nc, hyp, balance=(4.0, 1.0, 0.4)):
    """preds: list of [B,na,ny,nx,5+nc] raw Detect outputs; returns (loss [1], (box, obj, cls) [3]) as the reference's
    ComputeObjectLoss.__call__ (loss_functions.py:119-176) with BCEWithLogits(pos_weight), label smoothing, gr = 1."""
    eps = hyp.get("label_smoothing", 0.0)
    pos, neg = 1.0 - 0.5 * eps, 0.5 * eps
    shapes = [(p.shape[2], p.shape[3]) for p in preds]
    ents = assign_targets(labels, anchors, shapes, hyp["anchor_t"])
    box_term = obj_term = cls_term = torch.zeros(1)
    for l, (p, e) in enumerate(zip(preds, ents)):
        objectness_target = torch.zeros(p.shape[:4], dtype=p.dtype)
        n = e["b"].shape[0]
        if n:
            sel = p[e["b"], e["a"], e["gj"], e["gi"]]
            xy = sel[:, 0:2].sigmoid() * 2 - 0.5
            wh = (sel[:, 2:4].sigmoid() * 2) ** 2 * e["anchor"]
            quality = ciou(torch.cat((xy, wh), 1), e["box"])
            box_term = box_term + (1.0 - quality).mean()
            objectness_target[e["b"], e["a"], e["gj"], e["gi"]] = quality.detach().clamp(0).to(p.dtype)      # sequential: the last claim of a cell wins
            if nc > 1:
                want = torch.full_like(sel[:, 5:], neg)
                want[torch.arange(n), e["cls"]] = pos
                cls_term = cls_term + F.binary_cross_entropy_with_logits(sel[:, 5:], want, pos_weight=torch.tensor([hyp["cls_pw"]]))
        obj_term = obj_term + balance[l] * F.binary_cross_entropy_with_logits(p[..., 4], objectness_target, pos_weight=torch.tensor([hyp["obj_pw"]]))
    box_term, obj_term, cls_term = box_term * hyp["box"], obj_term * hyp["obj"], cls_term * hyp["cls"]
    return box_term + obj_term + cls_term, torch.cat((box_term, obj_term, cls_term)).detach()
