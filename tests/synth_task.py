"""TEST INFRASTRUCTURE (imports oracle/).  Brief training on a learnable synthetic task, then the accuracy-parity evaluation of
SURVEY.md 8(d):

    python tests/synth_task.py [--version s] [--size 320] [--steps 400] [--batch 16]

Task ("shapes"): bright rectangles of three colour classes on smooth noise; ground truth = the rectangles (YOLO boxes) and their
corners (keypoints).  The model is constructed by the reference-API constructor (its own initialisation), trained with
yolopoint_amd.engine.TrainStep (bf16, native forward / backward / loss kernels), and the resulting checkpoint is evaluated twice on
fresh images: HIP path (f16 forward, HIP box NMS) and CPU oracle (fp32 forward, oracle NMS), both scored with the reference's
metric definitions (oracle/eval_oracle.py) against the same ground truth.  Prints one JSON line; tests/test_gpu_accuracy_parity.py
runs the same functions with an assertion on |mAP difference| <= 0.2 pt.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def shapes_batch(B, S, device, seed, max_rects=5):
    """Images [B,3,S,S] in [0,1], YOLO labels [M,6] (image, class, xc, yc, w, h normalised), corner keypoint map [B,1,S,S].
    Generated with batched tensor ops on `device` from a seeded CPU parameter draw (identical on every device)."""
    g = torch.Generator().manual_seed(seed)
    K = max_rects
    low = torch.rand(B, 3, S // 16, S // 16, generator=g)
    n = torch.randint(2, K + 1, (B,), generator=g)
    wh = ((torch.rand(B, K, 2, generator=g) * 0.27 + 0.08) * S).floor()
    xy0 = (torch.rand(B, K, 2, generator=g) * (S - wh - 16) + 8).floor()
    cls = torch.randint(0, 3, (B, K), generator=g)
    nseed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g))
    active = torch.arange(K)[None, :] < n[:, None]
    dev = torch.device(device)
    low, wh, xy0, cls, active = (t.to(dev) for t in (low, wh, xy0, cls, active))
    gd = torch.Generator(device=dev).manual_seed(nseed)
    img = torch.nn.functional.interpolate(low, size=(S, S), mode="bilinear", align_corners=False) * 0.35 + 0.1
    img = img + torch.rand(B, 3, S, S, generator=gd, device=dev) * 0.05
    ys, xs = torch.arange(S, device=dev)[None, :, None], torch.arange(S, device=dev)[None, None, :]
    x1y1 = xy0 + wh
    Hc = S // 8
    has = torch.zeros(B, Hc, Hc, dtype=torch.bool, device=dev)
    pos = torch.zeros(B, Hc, Hc, dtype=torch.long, device=dev)
    bi = torch.arange(B, device=dev)
    for k in range(K):                        # later rectangles paint over earlier ones
        inside = ((xs >= xy0[:, k, 0, None, None]) & (xs < x1y1[:, k, 0, None, None]) & (ys >= xy0[:, k, 1, None, None]) & (ys < x1y1[:, k, 1, None, None])
                  & active[:, k, None, None])                                              # [B,S,S]
        col = torch.full((B, 3), 0.15, device=dev)
        col[bi, cls[:, k]] = 0.9
        tex = col[:, :, None, None] + torch.rand(B, 3, S, S, generator=gd, device=dev) * 0.05
        img = torch.where(inside[:, None], tex, img)
        for cx, cy in ((xy0[:, k, 0], xy0[:, k, 1]), (x1y1[:, k, 0] - 1, xy0[:, k, 1]), (xy0[:, k, 0], x1y1[:, k, 1] - 1), (x1y1[:, k, 0] - 1, x1y1[:, k, 1] - 1)):
            cx, cy = cx.long(), cy.long()
            a = active[:, k]
            has[bi[a], cy[a] // 8, cx[a] // 8] = True
            pos[bi[a], cy[a] // 8, cx[a] // 8] = (cy[a] % 8) * 8 + cx[a] % 8
    cell = torch.nn.functional.one_hot(pos, 64).float() * has[..., None]
    kp = cell.view(B, Hc, Hc, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, 1, S, S)
    mask = torch.zeros(B, 1, S, S, device=dev)
    mask[:, :, 4:-4, 4:-4] = 1
    wimg = (img + torch.randn(B, 3, S, S, generator=gd, device=dev) * 0.02).clamp(0, 1)
    bsel, ksel = torch.nonzero(active, as_tuple=True)
    c = (xy0[bsel, ksel] + wh[bsel, ksel] / 2) / S
    labels = torch.cat((bsel[:, None].float(), cls[bsel, ksel][:, None].float(), c, wh[bsel, ksel] / S), 1)
    return dict(image=img.clamp(0, 1), warped_image=wimg, labels_2D=kp, warped_labels=kp.clone(), valid_mask=mask, warped_valid_mask=mask.clone(),
                box_labels=labels, inv_homographies=torch.eye(3, device=dev).repeat(B, 1, 1))


def train(version, S, steps, B, device, seed=0, lr=2e-3, log=None):
    from helpers import NAMES80
    from yolopoint_amd import models
    from yolopoint_amd.engine import TrainStep
    torch.manual_seed(seed)
    m = models.Model(names=NAMES80, model_name="YOLOPoint", version=version)
    m.set_compute_dtype("bf16")
    m = m.to(device).train()
    step = TrainStep(m, device, img_size=S, lr=lr)
    step.sparse = dict(num_samples_per_image=min(600, (S // 8) ** 2 // 2), num_masked_non_matches_per_match=100)
    for i in range(steps):
        batch = shapes_batch(B, S, device, 1000 + i)
        loss = step(batch)
        if log and (i % 250 == 0 or i == steps - 1):
            with torch.no_grad():
                o = m(batch["image"])
                items = step.obj_loss(o["objects"], batch["box_labels"])[1]
                mx = max(float(t[..., 4].sigmoid().max()) for t in o["objects"])
            log(f"  step {i:4d} loss {float(loss):.4f}  obj-loss items (box, obj, cls) {[round(float(v), 4) for v in items]}  max objectness {mx:.3f}")
    return m


def evaluate(m, version, S, n_images, device, conf=0.001, iou=0.6, seed=50_000, chunk=16, kp_thresh=0.015, kp_nms=4):
    """mAP (IoU .5 and .5:.95, 101-point, reference metric definitions) and keypoint repeatability (k = 300, 3 px) of the CPU oracle and
    of the HIP path on the same fresh images against the same ground truth -> dict(map50_*, map_*, rep_*, n_*)."""
    from oracle import net_oracle, postproc_oracle as po, eval_oracle as eo
    from yolopoint_amd.utils import utils as U
    from yolopoint_amd.utils.general_yolo import non_max_suppression
    from yolopoint_amd.utils.loss_functions import warp_image_batch
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    m.set_compute_dtype("f16")
    m = m.to(device).eval()
    iouv = np.linspace(0.5, 0.95, 10).astype(np.float32)
    th = 0.05
    Hn = torch.tensor([[np.cos(th), -np.sin(th), 0.04], [np.sin(th), np.cos(th), -0.03], [0.02, 0.01, 1.0]], dtype=torch.float32)
    Hinv = torch.linalg.inv(Hn)
    stats, reps, npts = {"cpu": [], "hip": []}, {"cpu": [], "hip": []}, {"cpu": 0, "hip": 0}
    for c in range(0, n_images, chunk):
        batch = shapes_batch(chunk, S, device, seed + c)
        img, lab = batch["image"].cpu(), batch["box_labels"].cpu().numpy()
        wimg = warp_image_batch(img, Hinv.repeat(chunk, 1, 1), mode="bilinear").contiguous()
        with torch.no_grad():
            ref, refw = net_oracle.yolopoint_forward(sd, img, version), net_oracle.yolopoint_forward(sd, wimg, version)
            got, gotw = m(img.to(device)), m(wimg.to(device))
        det_cpu = po.non_max_suppression(ref["objects"][0].numpy(), conf, iou, agnostic=False, multi_label=True, max_det=300)
        det_hip = [d.cpu().numpy() for d in non_max_suppression(got["objects"][0], conf, iou, labels=[], multi_label=True, agnostic=False, max_det=300)]
        for b in range(chunk):
            gt = lab[lab[:, 0] == b]
            xyxy = np.concatenate((gt[:, 2:4] - gt[:, 4:6] / 2, gt[:, 2:4] + gt[:, 4:6] / 2), 1) * S
            labels = np.concatenate((gt[:, 1:2], xyxy), 1).astype(np.float32)
            for name, dets in (("cpu", det_cpu[b]), ("hip", det_hip[b])):
                tp = eo.process_batch(dets, labels, iouv) if len(dets) else np.zeros((0, 10), bool)
                stats[name].append((tp, dets[:, 4], dets[:, 5], labels[:, 0]))
            data = dict(image=np.zeros((3, S, S), np.float32), homography=Hn.numpy(), inv_homography=Hinv.numpy())
            p1 = po.get_pts_from_semi(ref["semi"][b].numpy(), kp_thresh, kp_nms).T
            p2 = po.get_pts_from_semi(refw["semi"][b].numpy(), kp_thresh, kp_nms).T
            q1 = U.getPtsFromSemi(got["semi"][b], kp_thresh, kp_nms).T
            q2 = U.getPtsFromSemi(gotw["semi"][b], kp_thresh, kp_nms).T
            reps["cpu"].append(eo.compute_repeatability(dict(data, prob=p1, warped_prob=p2))[0])
            reps["hip"].append(eo.compute_repeatability(dict(data, prob=q1, warped_prob=q2))[0])
            npts["cpu"] += len(p1); npts["hip"] += len(q1)
    out = {}
    for name, st in stats.items():
        tp, cf, pc, tc = (np.concatenate(x) for x in zip(*st))
        ap, _ = eo.ap_per_class(tp, cf, pc, tc)
        out[f"map50_{name}"], out[f"map_{name}"], out[f"n_{name}"] = 100 * float(ap[:, 0].mean()), 100 * float(ap.mean()), int(len(cf))
        out[f"rep_{name}"], out[f"kpts_{name}"] = 100 * float(np.mean(reps[name])), npts[name]
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--version", default="s")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--images", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    m = train(a.version, a.size, a.steps, a.batch, dev, lr=a.lr, log=lambda s: print(s, flush=True))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    r = evaluate(m, a.version, a.size, a.images, dev)
    r.update(train_s=round(t1 - t0, 1), eval_s=round(time.perf_counter() - t1, 1), version=a.version, size=a.size, steps=a.steps, batch=a.batch)
    print(json.dumps(r), flush=True)
