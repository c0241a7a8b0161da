"""numpy restatement of the reference's post-processing.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py): never imported by the product path.

Index-selection work (keypoint grid NMS, box NMS, mutual nearest neighbours) is restated as the
plain sequential algorithm; the HIP kernels must reproduce its selections bit for bit.  Sort
ties, which the reference leaves to unstable sorts (np.argsort / Tensor.argsort), are resolved
here as "higher score first, then lower index" — the golden vectors use distinct scores.

Pinned against outputs of the imported reference: tests/golden/make_golden.py,
tests/test_oracle_golden.py.  Citations are relative to /root/reference/src.
"""
import numpy as np


# ------------------------------------------------------------------------------------------
# keypoint heat map
# ------------------------------------------------------------------------------------------
def depth_to_space8(cells):
    """[64,Hc,Wc] -> [8Hc,8Wc]; channel c lands at pixel (8h + c//8, 8w + c%8) (nn.PixelShuffle(8))."""
    _, Hc, Wc = cells.shape
    return cells.reshape(8, 8, Hc, Wc).transpose(2, 0, 3, 1).reshape(Hc * 8, Wc * 8)


def flatten_detection(semi):
    """utils/utils.py:232-262: softmax over 65 channels, drop the dustbin, depth-to-space.
    semi [65,Hc,Wc] -> [1,H,W]; [B,65,Hc,Wc] -> [B,1,H,W]   (float32)."""
    semi = np.asarray(semi, dtype=np.float32)
    batch = semi.ndim == 4
    s = semi if batch else semi[None]
    e = np.exp(s - s.max(axis=1, keepdims=True))
    dense = e / e.sum(axis=1, keepdims=True)
    heat = np.stack([depth_to_space8(d[:-1]) for d in dense])[:, None]
    return heat if batch else heat[0]


def flatten_detection_demo(semi):
    """demo.py:140-150: exp(x) / (sum + 1e-5) without max subtraction.  semi [65,Hc,Wc] -> [H,W]."""
    semi = np.asarray(semi, dtype=np.float32)
    dense = np.exp(semi)
    dense = dense / (np.sum(dense, axis=0) + np.float32(.00001))
    return depth_to_space8(dense[:-1])


def nms_fast(in_corners, H, W, dist_thresh):
    """utils/utils.py:118-182.  Visit corners by descending confidence on the rounded grid; a corner
    whose pixel is still pending is kept and clears the (2r+1)^2 window around it.  Returns
    (3xN' survivors sorted by conf desc, their indices into in_corners)."""
    in_corners = np.asarray(in_corners)
    order = np.argsort(-in_corners[2, :], kind="stable")
    corners = in_corners[:, order]
    rc = corners[:2, :].round().astype(int)
    n = rc.shape[1]
    if n == 0:
        return np.zeros((3, 0)).astype(int), np.zeros(0).astype(int)
    if n == 1:
        return np.vstack((rc, in_corners[2])).reshape(3, 1), np.zeros((1)).astype(int)
    r = dist_thresh
    PENDING, KEPT = 1, -1
    state = np.zeros((H + 2 * r, W + 2 * r), dtype=int)
    owner = np.zeros((H, W), dtype=int)
    for i in range(n):                      # later (lower-confidence) duplicates overwrite the owner
        state[rc[1, i] + r, rc[0, i] + r] = PENDING
        owner[rc[1, i], rc[0, i]] = i
    for i in range(n):
        x, y = rc[0, i] + r, rc[1, i] + r
        if state[y, x] == PENDING:
            state[y - r:y + r + 1, x - r:x + r + 1] = 0
            state[y, x] = KEPT
    ky, kx = np.where(state == KEPT)
    keep = owner[ky - r, kx - r]
    out = corners[:, keep]
    o2 = np.argsort(-out[-1, :], kind="stable")
    return out[:, o2], order[keep[o2]]


def get_pts_from_heatmap(heatmap, conf_thresh, nms_dist, border=4):
    """utils/utils.py:465-485: threshold (>=) -> nms_fast -> sort desc -> drop the 4-px border."""
    heatmap = np.asarray(heatmap)
    H, W = heatmap.shape
    ys, xs = np.where(heatmap >= conf_thresh)
    if len(ys) == 0:
        return np.zeros((3, 0))
    pts = np.zeros((3, len(ys)))
    pts[0], pts[1], pts[2] = xs, ys, heatmap[ys, xs]
    pts, _ = nms_fast(pts, H, W, nms_dist)
    pts = pts[:, np.argsort(-pts[2, :], kind="stable")]
    bad = (pts[0] < border) | (pts[0] >= W - border) | (pts[1] < border) | (pts[1] >= H - border)
    return pts[:, ~bad]


def get_pts_from_semi(semi, conf_thresh=0.015, nms_dist=4):
    """utils/utils.py:94-101."""
    return get_pts_from_heatmap(np.squeeze(flatten_detection(semi)), conf_thresh, nms_dist)


def labels2d_to_3d(labels, cell=8, add_dustbin=True):
    """utils/utils.py:184-209: PixelUnshuffle(8) (+ dustbin, normalise).  [B,1,H,W] -> [B,65|64,Hc,Wc]."""
    labels = np.asarray(labels, dtype=np.float32)
    B, _, H, W = labels.shape
    Hc, Wc = H // cell, W // cell
    cells = labels.reshape(B, Hc, cell, Wc, cell).transpose(0, 2, 4, 1, 3).reshape(B, cell * cell, Hc, Wc)
    if add_dustbin:
        dust = 1 - cells.sum(axis=1)
        dust[dust < 1.] = 0
        cells = np.concatenate((cells, dust[:, None]), axis=1)
        cells = cells / cells.sum(axis=1, keepdims=True)
    return cells


def get_masks(mask_2d, cell=8):
    """utils/utils.py:103-116."""
    return np.prod(labels2d_to_3d(mask_2d, cell, add_dustbin=False), axis=1)


# ------------------------------------------------------------------------------------------
# box NMS
# ------------------------------------------------------------------------------------------
def xywh2xyxy(x):
    """utils/general_yolo.py:623-630 (fp32 arithmetic preserved)."""
    y = np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def nms_greedy(boxes, scores, iou_thres):
    """torchvision.ops.nms as pinned in SURVEY.md 8c (the package itself is not in the reference tree):
    visit boxes by descending score; a box is kept unless an earlier kept box has
    inter / (area_i + area_j - inter) > iou_thres, all in fp32."""
    boxes = np.asarray(boxes, dtype=np.float32)
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    dead = np.zeros(len(boxes), dtype=bool)
    keep = []
    thr = np.float32(iou_thres)
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (area[i] + area[rest] - inter)
        dead[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False, multi_label=False, max_det=300, classes=None, labels=()):
    """utils/general_yolo.py:124-235 for nm=0 (`classes`: :199-200; a-priori `labels`: :171-178).
    prediction [B,N,5+nc] fp32 -> list of B arrays [n,6] (x1,y1,x2,y2,conf,cls)."""
    prediction = np.asarray(prediction, dtype=np.float32)
    nc = prediction.shape[2] - 5
    max_wh, max_nms = np.float32(7680), 30000
    multi_label = multi_label and nc > 1
    ct = np.float32(conf_thres)
    out = []
    for xi, x in enumerate(prediction):
        x = x[x[:, 4] > ct].copy()
        if labels and len(labels[xi]):
            lb = np.asarray(labels[xi], dtype=np.float32)
            v = np.zeros((len(lb), nc + 5), dtype=np.float32)
            v[:, :4], v[:, 4] = lb[:, 1:5], 1.0
            v[np.arange(len(lb)), lb[:, 0].astype(np.int64) + 5] = 1.0
            x = np.concatenate((x, v), 0)
        if not x.shape[0]:
            out.append(np.zeros((0, 6), dtype=np.float32))
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        if multi_label:
            i, j = np.nonzero(x[:, 5:] > ct)
            x = np.concatenate((box[i], x[i, 5 + j, None], j[:, None].astype(np.float32)), 1)
        else:
            j = x[:, 5:].argmax(1)
            conf = x[np.arange(len(x)), 5 + j]
            x = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32)), 1)[conf > ct]
        if classes is not None:
            x = x[np.isin(x[:, 5], np.asarray(classes, dtype=np.float32))]
        if not x.shape[0]:
            out.append(np.zeros((0, 6), dtype=np.float32))
            continue
        x = x[np.argsort(-x[:, 4], kind="stable")[:max_nms]]
        c = x[:, 5:6] * (np.float32(0) if agnostic else max_wh)
        keep = nms_greedy(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        out.append(x[keep])
    return out


# ------------------------------------------------------------------------------------------
# descriptors
# ------------------------------------------------------------------------------------------
def sample_desc_from_points(coarse_desc, pts, cell=8):
    """evaluations/descriptor_evaluation.py:148-181 (== demo.py:200-215): bilinear grid_sample with
    align_corners=True at coordinates normalised by the FULL resolution (x/(W/2)-1), zero padding,
    then per-point L2 normalisation.  coarse_desc [D,Hc,Wc] fp32, pts [>=2,N] -> [D,N] fp32."""
    d = np.asarray(coarse_desc, dtype=np.float32)
    if d.ndim == 4:
        d = d[0]
    D, Hc, Wc = d.shape
    pts = np.asarray(pts)
    if pts.ndim != 2 or pts.shape[1] == 0:
        return np.empty((D, 0))
    W, H = float(Wc * cell), float(Hc * cell)
    gx = (pts[0].astype(np.float64) / (W / 2.) - 1.).astype(np.float32)
    gy = (pts[1].astype(np.float64) / (H / 2.) - 1.).astype(np.float32)
    ix = ((gx + np.float32(1)) / np.float32(2)) * np.float32(Wc - 1)
    iy = ((gy + np.float32(1)) / np.float32(2)) * np.float32(Hc - 1)
    x0, y0 = np.floor(ix).astype(int), np.floor(iy).astype(int)
    x1, y1 = x0 + 1, y0 + 1
    out = np.zeros((D, pts.shape[1]), dtype=np.float32)
    for (xx, yy, wgt) in ((x0, y0, (x1 - ix) * (y1 - iy)), (x1, y0, (ix - x0) * (y1 - iy)),
                          (x0, y1, (x1 - ix) * (iy - y0)), (x1, y1, (ix - x0) * (iy - y0))):
        ok = (xx >= 0) & (xx < Wc) & (yy >= 0) & (yy < Hc)
        out[:, ok] += d[:, yy[ok], xx[ok]] * wgt[ok].astype(np.float32)
    return out / np.linalg.norm(out, axis=0)[None, :]


def nn_match_two_way(desc1, desc2, nn_thresh):
    """models/model_wrap.py:434-476 (== demo.py:300-341).  desc [D,N] unit columns ->
    [3,L] rows (idx1, idx2, distance): nearest neighbour both ways and distance < nn_thresh."""
    assert desc1.shape[0] == desc2.shape[0]
    if desc1.shape[1] == 0 or desc2.shape[1] == 0:
        return np.zeros((3, 0))
    assert nn_thresh > 0.0
    dmat = np.sqrt(2 - 2 * np.clip(desc1.T @ desc2, -1, 1))
    fwd = dmat.argmin(axis=1)
    score = dmat[np.arange(dmat.shape[0]), fwd]
    back = dmat.argmin(axis=0)
    keep = (score < nn_thresh) & (np.arange(len(fwd)) == back[fwd])
    m = np.zeros((3, int(keep.sum())))
    m[0], m[1], m[2] = np.arange(desc1.shape[1])[keep], fwd[keep], score[keep]
    return m


def filter_points(boxes, pts, H, W):
    """reference demo.py:176-196: keypoints [3,N] inside any rint(xyxy) box are dropped (numpy slice painting of a mask)."""
    mask = np.ones((H, W))
    for x0, y0, x1, y1 in np.rint(np.asarray(boxes)[:, :4]).astype(int):
        mask[y0:y1, x0:x1] = 0
    p = pts.transpose()
    keep = mask[p[:, 1].astype(int), p[:, 0].astype(int)] == 1
    return p[keep].transpose()


def frontend_postprocess(semi, coarse_desc, pred, det_thresh=0.015, nms=4, border=4, conf=0.25, iou=0.45, max_det=300, filter_pts=True):
    """The post-processing chain of reference demo.py:136-215 on given head outputs (semi [65,Hc,Wc], coarse_desc [D,Hc,Wc],
    pred [N,5+nc]; numpy) -> (pts [3,N], desc [D,N], boxes [n,6])."""
    heat = flatten_detection_demo(semi)
    H, W = heat.shape
    pts = get_pts_from_heatmap(heat, det_thresh, nms, border)
    boxes = non_max_suppression(pred[None], conf, iou, agnostic=True, multi_label=True, max_det=max_det)[0]
    if filter_pts and pts.shape[1]:
        pts = filter_points(boxes, pts, H, W)
    desc = sample_desc_from_points(coarse_desc, pts) if pts.shape[1] else np.zeros((coarse_desc.shape[0], 0))
    return pts, desc, boxes


# ------------------------------------------------------------------------------------------
# homography adaptation (export_homography.py:88-150)
# ------------------------------------------------------------------------------------------
def _linspace_m1_p1(n):
    """torch.linspace(-1, 1, n) in float32: filled symmetrically from both ends."""
    i = np.arange(n)
    step = np.float32(2.0) / np.float32(max(n - 1, 1))
    lo = np.float32(-1.0) + step * i.astype(np.float32)
    hi = np.float32(1.0) - step * (n - 1 - i).astype(np.float32)
    return np.where(i < n // 2, lo, hi).astype(np.float32)


def warp_image_batch(img, mat_homo_inv, mode="bilinear"):
    """utils/utils.py:333-376 for img [B,C,H,W]: every output pixel's normalised coordinate goes through mat_homo_inv[b]
    (warp_points :274-295), then F.grid_sample(align_corners=True, padding_mode='zeros').  float32 throughout."""
    img = np.asarray(img, dtype=np.float32)
    Hm = np.asarray(mat_homo_inv, dtype=np.float32).reshape(-1, 3, 3)
    B, C, H, W = img.shape
    gx, gy = np.meshgrid(_linspace_m1_p1(W), _linspace_m1_p1(H))                   # [H,W] each
    out = np.zeros_like(img)
    for b in range(B):
        h = Hm[b]
        X = h[0, 0] * gx + h[0, 1] * gy + h[0, 2]
        Y = h[1, 0] * gx + h[1, 1] * gy + h[1, 2]
        Z = h[2, 0] * gx + h[2, 1] * gy + h[2, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            ix = ((X / Z + np.float32(1)) / np.float32(2)) * np.float32(W - 1)
            iy = ((Y / Z + np.float32(1)) / np.float32(2)) * np.float32(H - 1)
        if mode == "nearest":
            xn, yn = np.rint(ix), np.rint(iy)
            ok = (xn >= 0) & (xn < W) & (yn >= 0) & (yn < H)
            xi, yi = np.where(ok, xn, 0).astype(np.int64), np.where(ok, yn, 0).astype(np.int64)
            out[b] = np.where(ok[None], img[b][:, yi, xi], 0)
            continue
        x0, y0 = np.floor(ix), np.floor(iy)
        acc = np.zeros((C, H, W), dtype=np.float32)
        for dy in (0, 1):
            for dx in (0, 1):
                xt, yt = x0 + dx, y0 + dy
                wx = (x0 + 1 - ix) if dx == 0 else (ix - x0)
                wy = (y0 + 1 - iy) if dy == 0 else (iy - y0)
                ok = (xt >= 0) & (xt < W) & (yt >= 0) & (yt < H)
                xi, yi = np.where(ok, xt, 0).astype(np.int64), np.where(ok, yt, 0).astype(np.int64)
                acc += np.where(ok[None], img[b][:, yi, xi] * (wx * wy).astype(np.float32)[None], np.float32(0))
        out[b] = acc
    return out


def homography_adaptation(semi, valid_mask, inv_homographies, conf_thresh=0.015, nms_dist=4, top_k=None):
    """export_homography.py:88-150 for one image: semi [N,65,Hc,Wc] of its N views, valid_mask [N,1,H,W], inv_homographies
    [N,3,3] -> (aggregated heat map [H,W], pts [n,3] (x, y, prob))."""
    heat = flatten_detection(semi) * np.asarray(valid_mask, dtype=np.float32)
    heat = warp_image_batch(heat, inv_homographies)
    mask = warp_image_batch(valid_mask, inv_homographies)
    with np.errstate(divide="ignore", invalid="ignore"):
        agg = heat.sum(axis=0, dtype=np.float32) / mask.sum(axis=0, dtype=np.float32)
    agg = agg[0]
    pts = get_pts_from_heatmap(agg, conf_thresh, nms_dist).transpose()
    if top_k and pts.shape[0] > top_k:
        pts = pts[:top_k, :]
    return agg, pts


# ------------------------------------------------------------------------------------------
# PointTracker bookkeeping (models/model_wrap.py:410-606)
# ------------------------------------------------------------------------------------------
class PointTrackerOracle:
    """Sequential restatement of the reference tracker: one match at a time, as the reference loops."""

    def __init__(self, max_length=2, nn_thresh=0.7):
        assert max_length >= 2
        self.maxl, self.nn_thresh = max_length, nn_thresh
        self.all_pts = [np.zeros((2, 0)) for _ in range(max_length)]
        self.last_desc, self.last_pts, self.matches = None, None, None
        self.tracks = np.zeros((0, max_length + 2))
        self.track_count, self.max_score = 0, 9999

    def get_offsets(self):                                    # :477-491
        sizes = [0] + [self.all_pts[i].shape[1] for i in range(len(self.all_pts) - 1)]
        return np.cumsum(np.array(sizes))

    def update(self, pts, desc):                              # :503-577
        if self.last_desc is None:
            self.last_desc = np.zeros((desc.shape[0], 0))
        gone = self.all_pts.pop(0).shape[1]
        self.all_pts.append(pts)
        self.tracks = np.delete(self.tracks, 2, axis=1)
        for col in range(2, self.tracks.shape[1]):
            self.tracks[:, col] -= gone
        self.tracks[:, 2:][self.tracks[:, 2:] < -1] = -1
        offsets = self.get_offsets()
        self.tracks = np.hstack((self.tracks, -1 * np.ones((self.tracks.shape[0], 1))))
        taken = np.zeros(pts.shape[1], dtype=bool)
        matches = nn_match_two_way(self.last_desc, desc, self.nn_thresh)
        self.matches = matches
        if self.last_pts is not None:
            self.matches = np.concatenate((self.last_pts[:, matches[0, :].astype(int)], pts[:2, matches[1, :].astype(int)]), axis=0)
        for m in matches.T:
            a, b = int(m[0]) + offsets[-2], int(m[1]) + offsets[-1]
            rows = np.argwhere(self.tracks[:, -2] == a)
            if rows.shape[0] == 0:
                continue
            taken[int(m[1])] = True
            r = int(rows[0, 0])
            self.tracks[r, -1] = b
            if self.tracks[r, 1] == self.max_score:
                self.tracks[r, 1] = m[2]
            else:
                n_obs = (self.tracks[r, 2:] != -1).sum() - 1.0
                self.tracks[r, 1] = (1.0 - 1.0 / n_obs) * self.tracks[r, 1] + (1.0 / n_obs) * m[2]
        ids = (np.arange(pts.shape[1]) + offsets[-1])[~taken]
        fresh = -1 * np.ones((ids.shape[0], self.maxl + 2))
        fresh[:, -1] = ids
        fresh[:, 0] = self.track_count + np.arange(ids.shape[0])
        fresh[:, 1] = self.max_score
        self.tracks = np.vstack((self.tracks, fresh))
        self.track_count += ids.shape[0]
        self.tracks = self.tracks[np.any(self.tracks[:, 2:] >= 0, axis=1), :]
        self.last_desc, self.last_pts = desc.copy(), pts[:2, :].copy()

    def get_tracks(self, min_length):                         # :579-596
        long_enough = np.sum(self.tracks[:, 2:] != -1, axis=1) >= min_length
        return self.tracks[long_enough & (self.tracks[:, -1] != -1), :].copy()
