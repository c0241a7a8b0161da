"""BASELINE.json: "det mAP / kp repeatability parity ... within 0.2 pt on the same synthetic eval".

Identical seeded weights go through (a) the CPU oracle end to end (fp32 forward, oracle keypoint/box post-processing) and
(b) the HIP path end to end (f16 forward, HIP keypoint decode/NMS and box NMS); both are scored with the reference's metric
definitions (oracle/eval_oracle.py) against the same ground truth on the same synthetic image pairs."""
import numpy as np
import pytest
import torch

from helpers import make_model
from oracle import net_oracle, postproc_oracle as po, eval_oracle as eo
from yolopoint_amd.utils import utils as U
from yolopoint_amd.utils.general_yolo import non_max_suppression
from yolopoint_amd.utils.loss_functions import warp_image_batch

pytestmark = pytest.mark.gpu


def smooth_images(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, S // 8, S // 8, generator=g)
    x = torch.nn.functional.interpolate(x, size=(S, S), mode="bicubic", align_corners=False).clamp(0, 1)
    return (0.7 * x + 0.3 * torch.rand(B, 3, S, S, generator=g)).contiguous()


@pytest.mark.parametrize("dtype,twin_iou,twin_conf,twin_frac", [("f32", 0.002, 0.001, 0.99), ("f16", 0.05, 0.05, 0.90)])
def test_repeatability_and_map_parity(cuda, dtype, twin_iou, twin_conf, twin_frac):
    version, B, S = "s", 8, 256
    m, sd = make_model(version, 77, dtype=dtype)
    # the seeded synthetic checkpoint has head logits of magnitude ~40 (a trained network: a few units), which would turn
    # 16-bit relative rounding into several-percent box-size errors through (2*sigmoid)^2; bring the Detect logits to O(4)
    for i in range(3):
        sd[f"model.Detect.m.{i}.weight"] = sd[f"model.Detect.m.{i}.weight"] * 0.1
    m.load_state_dict(sd, strict=True)
    m = m.to(cuda)
    img = smooth_images(B, S, 5)
    th = 0.04
    Hn = torch.tensor([[np.cos(th), -np.sin(th), 0.03], [np.sin(th), np.cos(th), -0.02], [0.015, 0.01, 1.0]], dtype=torch.float32)
    Hinv = torch.linalg.inv(Hn)
    wimg = warp_image_batch(img, Hinv.repeat(B, 1, 1), mode="bilinear").contiguous()
    with torch.no_grad():
        ref, refw = net_oracle.yolopoint_forward(sd, img, version), net_oracle.yolopoint_forward(sd, wimg, version)
        got, gotw = m(img.to(cuda)), m(wimg.to(cuda))
    thr, r = 0.015, 4
    reps = {"cpu": [], "hip": []}
    for b in range(B):
        data = dict(image=np.zeros((3, S, S), np.float32), homography=Hn.numpy(), inv_homography=Hinv.numpy())
        p1 = po.get_pts_from_semi(ref["semi"][b].numpy(), thr, r).T
        p2 = po.get_pts_from_semi(refw["semi"][b].numpy(), thr, r).T
        reps["cpu"].append(eo.compute_repeatability(dict(data, prob=p1, warped_prob=p2))[0])
        q1 = U.getPtsFromSemi(got["semi"][b], thr, r).T
        q2 = U.getPtsFromSemi(gotw["semi"][b], thr, r).T
        reps["hip"].append(eo.compute_repeatability(dict(data, prob=q1, warped_prob=q2))[0])
        assert p1.shape[0] > 50 and q1.shape[0] > 50
    rep_cpu, rep_hip = 100 * float(np.mean(reps["cpu"])), 100 * float(np.mean(reps["hip"]))
    # ---- detection mAP against a common ground truth (the oracle's 12 most confident boxes per image)
    conf, iou = 0.7, 0.45
    det_cpu = po.non_max_suppression(ref["objects"][0].numpy(), conf, iou, agnostic=False, multi_label=False, max_det=2000)
    det_hip = [d.cpu().numpy() for d in non_max_suppression(got["objects"][0], conf, iou, labels=[], multi_label=False, agnostic=False, max_det=2000)]
    iouv = np.linspace(0.5, 0.95, 10).astype(np.float32)
    maps = {}
    # ground truth, built SYMMETRICALLY so that neither path is favoured: per image 60 boxes drawn (seeded) from the oracle's
    # detections and 60 from the HIP path's, each jittered.  (Random weights give dense, near-tied scores, so a ground truth
    # taken from one path alone would rank that path's detections perfectly and bias the comparison.)
    rng = np.random.default_rng(3)
    gts = []
    for b in range(B):
        picks = []
        for dets in (det_cpu[b], det_hip[b]):
            idx = rng.choice(min(len(dets), 200), size=min(60, len(dets)), replace=False)
            picks.append(dets[idx])
        gt = np.concatenate(picks).copy()
        wh = gt[:, 2:4] - gt[:, 0:2]
        c = (gt[:, 0:2] + gt[:, 2:4]) / 2 + rng.normal(0, 0.03, wh.shape) * wh
        wh = wh * rng.uniform(0.85, 1.15, wh.shape)
        gts.append(np.concatenate((gt[:, 5:6], c - wh / 2, c + wh / 2), 1).astype(np.float32))
    for name, dets in (("cpu", det_cpu), ("hip", det_hip)):
        tps, confs, pcls, tcls = [], [], [], []
        for b in range(B):
            labels = gts[b]
            tps.append(eo.process_batch(dets[b], labels, iouv)); confs.append(dets[b][:, 4]); pcls.append(dets[b][:, 5]); tcls.append(labels[:, 0])
        ap, _ = eo.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls))
        maps[name] = (100 * ap[:, 0].mean(), 100 * ap.mean())
    print(f"repeatability cpu {rep_cpu:.3f} hip {rep_hip:.3f} | mAP@.5 cpu {maps['cpu'][0]:.3f} hip {maps['hip'][0]:.3f} | "
          f"mAP@.5:.95 cpu {maps['cpu'][1]:.3f} hip {maps['hip'][1]:.3f} | detections {[len(d) for d in det_cpu]} vs {[len(d) for d in det_hip]}")
    assert sum(len(d) for d in det_cpu) > 40
    assert abs(rep_cpu - rep_hip) <= 0.2, (rep_cpu, rep_hip)
    # Detection parity is asserted on the detections themselves: with seeded random weights the score landscape is dense and
    # near-tied, so AP (which hinges on which of several overlapping boxes is the single best match of a label) moves by
    # points between two detection sets that agree box for box — it is printed above for the record, and the <= 0.2 pt mAP
    # bar is left to a trained checkpoint.  What is asserted: detection-level agreement — an oracle
    # detection has a HIP detection of the same class whose corners lie within `twin_iou` x box size (+0.5 px) and whose
    # confidence is within `twin_conf` (fp32 path: 0.2 % / 0.001 for 99 %; f16 path: 5 % / 0.05 for 90 %).  Exact NMS parity on
    # identical inputs is tests/test_gpu_postproc.py.  (Seeded random heads also emit zero-area boxes, so IoU is not used.)
    matched = total = 0
    for dc, dh in zip(det_cpu, det_hip):
        size = np.maximum(dc[:, 2:4] - dc[:, 0:2], 1.0).max(1)                                     # [N]
        dcorner = np.abs(dc[:, None, :4] - dh[None, :, :4]).max(2)                                  # [N, M]
        ok = (dcorner <= twin_iou * size[:, None] + 0.5) & (dc[:, 5:6] == dh[None, :, 5]) & (np.abs(dc[:, 4:5] - dh[None, :, 4]) <= twin_conf)
        matched += int(ok.any(1).sum()); total += len(dc)
    print(f"{dtype}: {matched}/{total} oracle detections have a HIP twin")
    assert matched / total >= twin_frac, (matched, total)
    assert abs(sum(len(d) for d in det_cpu) - sum(len(d) for d in det_hip)) <= 0.02 * total


@pytest.mark.statistical
def test_trained_checkpoint_map_and_repeatability_parity(cuda):
    """BASELINE.json "det mAP / kp repeatability parity within 0.2 pt on the same synthetic eval", on a checkpoint the build TRAINED
    itself: YOLOPoint-s, 500 bf16 optimizer steps of yolopoint_amd.engine.TrainStep on the synthetic shapes task
    (tests/synth_task.py: coloured rectangles = boxes of 3 classes, their corners = keypoints), then the same 64 fresh images go
    through (a) the CPU oracle in fp32 and (b) the HIP path in f16 with the HIP NMS / keypoint kernels; both are scored with the
    reference's metric definitions against the same ground truth."""
    import synth_task
    m = synth_task.train("s", 256, 500, 16, cuda, lr=2e-3)
    r = synth_task.evaluate(m, "s", 256, 64, cuda)
    print(r)
    assert r["map50_cpu"] > 60 and r["n_cpu"] > 200, r                 # the checkpoint detects: the comparison is not vacuous
    assert r["kpts_cpu"] > 200, r
    assert abs(r["map50_cpu"] - r["map50_hip"]) <= 0.2, r
    assert abs(r["map_cpu"] - r["map_hip"]) <= 0.2, r
    assert abs(r["rep_cpu"] - r["rep_hip"]) <= 0.2, r
