"""GPU-resident frame pipeline: the reference's `YoloPointFrontend.process_img` (src/demo.py:58-230) with every stage on the
device -- forward, keypoint decode (the demo's exp / (sum + 1e-5) softmax), threshold + greedy grid NMS + border removal,
box NMS, box-mask keypoint filtering, descriptor sampling -- and exactly one host synchronisation per frame (two counters).
The reference copies `semi` to numpy, runs the decode / NMS loops in Python and copies the points back for grid_sample.

    fe = YoloPointFrontend(model, device)
    pts, desc, boxes = fe.process_img(img_uint8_hwc)        # reference formats: [3,N] float64, [D,N] float32, [tensor[n,6]]
    d = fe.process_tensor(x)                                 # device tensors, for the tracker / the next stage
"""
import ctypes as C

import numpy as np
import torch

from . import _hip
from .utils._ws import workspace

SP_DEFAULT = dict(detection_threshold=0.015, nms=4)                       # reference configs/*_inference.yaml
YOLO_DEFAULT = dict(conf_thres_box=0.25, iou_thres_box=0.45, max_det=300)


class YoloPointFrontend:
    def __init__(self, model, device, sp_config=None, yolo_config=None, filter_pts=True, border_remove=4, cell=8, crop_resize=None, freeze_weights=True):
        if crop_resize:
            raise _hip.YpError("YoloPointFrontend: crop_resize needs cv2.resize and is not part of the device pipeline")
        if cell != 8:
            raise _hip.YpError("YoloPointFrontend: the decode kernel is specialised for cell = 8")
        self.model, self.device = model.eval(), torch.device(device)
        self.sp_config = dict(SP_DEFAULT, **(sp_config or {}))
        self.yolo_config = dict(YOLO_DEFAULT, **(yolo_config or {}))
        self.filter_pts, self.border_remove, self.cell = filter_pts, border_remove, cell
        # benchmarking aid: a list of (semi [1,65,Hc,Wc], pred [1,rows,no]) device tensors fed to the post-processing IN PLACE of the model's
        # keypoint / Detect outputs, frame i taking entry i % len (random-weight heads saturate: SURVEY.md 8d); the forward still runs in full
        self.planted, self._frame = None, 0
        # the keypoint post-processing hangs into the forward's side lane (see process_tensor) when the model is this package's YOLOPoint
        net = getattr(self.model, "model", None)
        self._net = net if hasattr(net, "_emit_heads_hook") else None
        # (both change the WRAPPED model for as long as this front end lives: its plans carry a callback op -- they replay eagerly, also for
        # other callers -- and, with freeze_weights, in-place edits / .to() / .half() of its parameters are not seen until
        # freeze_weights(False).  close() -- or leaving the `with` block -- puts the model back as it was.)
        self._restore = None
        if self._net is not None:
            self._restore = (bool(getattr(self._net, "heads_hook", False)), "_frozen_version" in self._net.__dict__)
            self._net.heads_hook = True
            if freeze_weights:
                self._net.freeze_weights()

    def close(self):
        """Give the wrapped model back: no heads hook in the plans it builds from now on, weights unfrozen unless they were frozen before."""
        if self._net is not None and self._restore is not None:
            hook, frozen = self._restore
            self._net.heads_hook = hook
            self._net.__dict__.pop("_heads_cb", None)
            if not frozen:
                self._net.freeze_weights(False)
            self._restore = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- demo.py:111-121: make both dims divisible by 32 by a centred crop
    @staticmethod
    def preprocess(img):
        h0, w0 = img.shape[:2]
        cut_h, cut_w = (h0 % 32) / 2, (w0 % 32) / 2
        cut_h0, cut_h1 = int(np.ceil(cut_h)), int(np.floor(cut_h))
        cut_w0, cut_w1 = int(np.ceil(cut_w)), int(np.floor(cut_w))
        return img[cut_h0:h0 - cut_h1, cut_w0:w0 - cut_w1], cut_h0, cut_w0, 1.0

    NMS_ROUNDS = 8        # fix-point rounds enqueued per frame by the non-synchronising keypoint NMS (planted 1280x1280 maps need ~6; a frame that
                          # needs more is redone with the converging variant below -- 21 us per round at 1280x1280)

    @_hip.guarded
    @torch.no_grad()
    def process_tensor(self, inp, sync_nms=False):
        """inp: float [1,3,H,W] on the device, values in [0,1].  Returns dict(pts [N,3] (x, y, conf; conf descending),
        desc [D,N] L2-normalised columns, boxes [n,6] (xyxy, conf, cls)) -- all device tensors."""
        if inp.dim() != 4 or inp.shape[0] != 1:
            raise _hip.YpError("YoloPointFrontend.process_tensor: one frame per call ([1,3,H,W])")
        l, st, dev = _hip.lib(), _hip.stream_ptr(), inp.device
        _, _, H, W = inp.shape
        Hc, Wc = H // 8, W // 8
        heat = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radius = int(self.sp_config["nms"])
        step = radius + 1
        max_pts = max(1, -(-H // step) * -(-W // step))
        pts = torch.empty((max_pts, 3), dtype=torch.float32, device=dev)
        counts = torch.zeros((5,), dtype=torch.int32, device=dev)   # [points after NMS, boxes, points after filtering, NMS candidates left undecided, pixels >= threshold]
        ws = workspace(dev, l.yp_kp_nms_workspace_bytes(1, H, W), "kp_nms")
        planted = self.planted[self._frame % len(self.planted)] if self.planted else None
        self._frame += 1

        def keypoints(semi, stream):
            """demo softmax -> heat map -> threshold, grid NMS, border removal, sort by confidence, on `stream`"""
            sb, sc, sy, sx = semi.stride()
            _hip.check(l.yp_kp_decode(semi.data_ptr(), 1, Hc, Wc, sb, sc, sy, sx, 1, heat.data_ptr(), stream))
            kp_args = (heat.data_ptr(), 1, H, W, float(self.sp_config["detection_threshold"]), radius, int(self.border_remove), pts.data_ptr(),
                       counts.data_ptr(), max_pts, ws.data_ptr(), ws.numel())
            if sync_nms:
                _hip.check(l.yp_kp_nms(*kp_args, stream))
            else:       # a fixed number of fix-point rounds, no host synchronisation inside; convergence is checked with the frame's counters
                _hip.check(l.yp_kp_nms_async(*kp_args, self.NMS_ROUNDS, counts[3:4].data_ptr(), stream))

        # The keypoint post-processing needs the keypoint head only.  A YOLOPoint plan with `heads_hook` calls back from its SIDE lane once the
        # heads are enqueued there (models/YOLOPoint.py::_emit_heads_hook): decode + NMS (~0.35 ms at 1280 x 1280) then run beside the YOLO
        # encoder / PAN chain of the forward instead of behind it; the plan's join orders them before everything below.
        net = self._net
        fired = []
        if net is not None and not sync_nms:
            def hook(stream):
                if planted is not None:
                    semi_ = planted[0]
                else:
                    v = next(iter(net._plans.values()))[2]["semi"]
                    semi_ = v.buf.t[..., :65].permute(0, 3, 1, 2)
                keypoints(semi_, C.c_void_p(stream))
                fired.append(True)
            net._heads_cb = hook
        try:
            outs = self.model(inp)
        finally:
            if net is not None:
                net._heads_cb = None
        semi, coarse, (pred, _) = outs["semi"], outs["desc"], outs["objects"]
        if planted is not None:
            semi, pred = planted
        assert tuple(semi.shape[-2:]) == (Hc, Wc)
        if not fired:
            keypoints(semi, st)
        # boxes: multi-label, class-agnostic NMS as the demo calls it (demo.py:168-174)
        p = pred if (pred.dtype == torch.float32 and pred.is_contiguous()) else pred.float().contiguous()
        N, no = p.shape[1], p.shape[2]
        nc, max_det = no - 5, int(self.yolo_config["max_det"])
        ml = int(nc > 1)
        boxes = torch.empty((1, max_det, 6), dtype=torch.float32, device=dev)
        ws2 = workspace(dev, l.yp_box_nms_workspace_bytes(1, N, nc, ml, 30000), "box_nms")
        cnt_boxes = counts[1:2]
        _hip.check(l.yp_box_nms(p.data_ptr(), 1, N, nc, float(self.yolo_config["conf_thres_box"]), float(self.yolo_config["iou_thres_box"]), ml, 1,
                                max_det, 30000, 7680.0, boxes.data_ptr(), cnt_boxes.data_ptr(), ws2.data_ptr(), ws2.numel(), st))
        # keypoints inside a detected box are dropped (dynamic objects)
        if self.filter_pts:
            kept = torch.empty_like(pts)
            _hip.check(l.yp_pts_box_filter(pts.data_ptr(), counts[0:1].data_ptr(), max_pts, boxes.data_ptr(), cnt_boxes.data_ptr(), max_det, 6, H, W,
                                           kept.data_ptr(), counts[2:3].data_ptr(), st))
        else:
            kept = pts
            counts[2:3].copy_(counts[0:1])
        # pixels that passed the threshold: the NMS kernels' candidate counter, where the library says it lives in their workspace
        off = l.yp_kp_nms_candidate_count_offset(1, H, W)
        counts[4:5].copy_(ws[off:off + 4].view(torch.int32))
        n_nms, n_box, n_pts, undecided, n_cand = counts.cpu().tolist()            # the frame's only host sync
        if undecided:                                                              # the greedy NMS needed more rounds than enqueued: redo with the
            self._frame -= 1                                                       # (same planted entry, if any)
            return self.process_tensor(inp, sync_nms=True)                         # converging variant (not seen on real or planted heat maps)
        if n_box < 0:
            raise _hip.YpError("YoloPointFrontend: box NMS candidate list overflowed its workspace")
        kept, boxes = kept[:n_pts], boxes[0, :n_box]
        D = coarse.shape[1]
        desc = torch.empty((D, n_pts), dtype=torch.float32, device=dev)
        if n_pts:
            xy = kept[:, :2].contiguous()
            _, dc, dy, dx = coarse.stride()
            _hip.check(l.yp_desc_sample(coarse.data_ptr(), D, Hc, Wc, dc, dy, dx, xy.data_ptr(), n_pts, 8, desc.data_ptr(), st))
        return {"pts": kept, "desc": desc, "boxes": boxes, "n_before_filter": n_nms, "n_candidates": n_cand, "heat": heat[0]}

    @torch.no_grad()
    def process_img(self, img):
        """img: HxWx3 uint8 (numpy).  Returns (pts [3,N] float64, desc [D,N] float32, [boxes tensor]) in the coordinates of the
        original image, or (zeros((3,0)), None, None) when no pixel reaches the detection threshold -- the reference's return
        convention (demo.py:151-153; a frame whose candidates are all removed later returns empty arrays and the boxes)."""
        img, cth, ctw, fac = self.preprocess(img)
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(self.device).float().div_(255.).unsqueeze(0)
        r = self.process_tensor(x)
        if r["n_candidates"] == 0:
            return np.zeros((3, 0)), None, None
        pts = r["pts"].cpu().numpy().astype(np.float64).T.copy()
        desc = r["desc"].cpu().numpy()
        pts[0] = (pts[0] + ctw) / fac
        pts[1] = (pts[1] + cth) / fac
        boxes = r["boxes"].clone()
        boxes[:, :4] = (boxes[:, :4] + torch.tensor([ctw, cth, ctw, cth], device=boxes.device)) / fac
        return pts, desc, [boxes]


def to_keypoint_array(pts, desc):
    """(pts [3,N] (x, y, conf), desc [D,N]) -> the fields of the reference's `KeypointArray` ROS message
    (src/ros_messages/keypoint_msg/msg/KeypointArray.msg; filled by yolopoint_ros.py:109-117): uint16[] x, uint16[] y,
    float32[] score, uint8 desc_len, float32[] desc_flat (the [D,N] matrix flattened row-major, i.e. dimension-major).
    Kept as the reference writes it: `x` carries pts[1] and `y` carries pts[0]; desc_len is D modulo 256 (the field is a uint8,
    so D = 256 reads 0).  Device tensors are converted on the device and cross to the host once."""
    if isinstance(pts, torch.Tensor) and isinstance(desc, torch.Tensor) and pts.is_cuda:
        n, D = pts.shape[1], desc.shape[0]
        blob = torch.cat((pts[1].to(torch.int32).to(torch.float32), pts[0].to(torch.int32).to(torch.float32), pts[2].float(), desc.float().flatten())).cpu().numpy()
        x, y, score, flat = blob[:n], blob[n:2 * n], blob[2 * n:3 * n], blob[3 * n:]
    else:
        pts, desc = np.asarray(pts), np.asarray(desc)
        D = desc.shape[0]
        x, y, score, flat = pts[1, :], pts[0, :], pts[2, :], desc.flatten()
    return {"x": x.astype(np.uint16), "y": y.astype(np.uint16), "score": score.astype(np.float32), "desc_len": np.uint8(D % 256),
            "desc_flat": np.ascontiguousarray(flat, dtype=np.float32)}
