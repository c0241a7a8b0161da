"""One optimizer step of YOLOPoint training as the reference runs it (src/train.py:189-259), on synthetic batches.

  outs   = model(image)           outs_w = model(warped_image)                (train.py:208,220)
  loss   = (det + det_warp) + lambda_loss * infonce + lambda_loss_obj * obj    (train.py:238-241)
  loss.backward(); [gradient all-reduce over ranks]; optimizer.step()         (train.py:245-252)

Forward/backward of the network run through the native plans (yolopoint_amd/training.py).  The loss stage has two formulations with
bit-identical gradients: the native one (TrainStep._loss_and_grads_native: loss kernels between the plans' own buffers, label
preparation and the loss sum as native launches -- no framework kernel in a steady-state step) and the autograd one over the
reference's loss API (utils/loss_functions.py), which YOLOPointv52 / two-graph mode use.  Adam: optim.FlatAdam (one launch) or
torch.optim.Adam (train.py:88).  Data: SURVEY.md 8(d) synthetic recipe, generated on the device.
"""
import collections
import contextlib
import ctypes as C
import os
from .switches import sw

import torch

from .utils.loss_functions import ComputeDetectorLoss, ComputeObjectLoss, infonce, infonce_prepare
from .utils.utils import labels2Dto3D, getMasks
from .dp import GradAllReducer

# reference configs/coco.yaml:113-149
HYP = dict(box=0.05, cls=0.5, obj=1.0, anchor_t=4.0, fl_gamma=0.0, cls_pw=1.0, obj_pw=1.0)
LAMBDA_DESC, LAMBDA_OBJ = 0.1, 10.0
SPARSE = dict(num_samples_per_image=3000, num_masked_non_matches_per_match=200)


def synthetic_batch(B, S, device, seed, nc=80):
    """image / warped image U[0,1); one keypoint label per 8x8 cell with prob 0.12; valid mask with a 4-px zero
    border; 8 boxes per image; identity homographies (throughput runs)."""
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *shape: torch.rand(*shape, generator=g, device=device)
    Hc = S // 8
    img, img_w = r(B, 3, S, S), r(B, 3, S, S)

    def labels():
        has = (r(B, Hc, Hc) < 0.12)
        pos = torch.randint(0, 64, (B, Hc, Hc), generator=g, device=device)
        cell = torch.nn.functional.one_hot(pos, 64).float() * has[..., None]                 # [B,Hc,Hc,64]
        return cell.view(B, Hc, Hc, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, 1, S, S)
    mask = torch.zeros(B, 1, S, S, device=device)
    mask[:, :, 4:-4, 4:-4] = 1
    nb = 8
    boxes = torch.cat((torch.arange(B, device=device).repeat_interleave(nb)[:, None].float(),
                       torch.randint(0, nc, (B * nb, 1), generator=g, device=device).float(),
                       0.1 + 0.8 * r(B * nb, 2), torch.exp(torch.log(torch.tensor(0.03)) + r(B * nb, 2) * (torch.log(torch.tensor(0.5 / 0.03))))), 1)
    eye = torch.eye(3, device=device).repeat(B, 1, 1)
    return dict(image=img, warped_image=img_w, labels_2D=labels(), warped_labels=labels(), valid_mask=mask, warped_valid_mask=mask.clone(),
                box_labels=boxes, inv_homographies=eye)


def _record_stream(obj, stream):
    """Tensors made on the side stream are consumed on `stream`: tell the caching allocator."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record_stream(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record_stream(o, stream)


class TrainStep:
    """One optimizer step of the reference loop (train.py:189-259) with its data-parallel and accumulation semantics:
    `gas` micro-batches per optimizer step (accelerator.accumulate: the loss of each is divided by gas, the gradient all-reduce runs
    on the last one only), optional gradient clipping (train.py:249-250) and LambdaLR schedule (train.py:91-93, stepped by the caller
    once per epoch: `step.scheduler`)."""

    def __init__(self, model, device, img_size=640, lr=1e-3, group=None, gas=1, max_grad_norm=None, lr_lambda=None, fp8=False):
        """fp8 (BASELINE.json configs[4]; the model's compute dtype must be bf16): the Conv layers with channel counts that are multiples
        of 64 multiply 8-bit operands (training.TrainGraph); the first call runs two extra forward / backward passes on its batch, without
        an optimizer step, to calibrate the per-tensor scales (they start at 1 and follow the previous step's maxima afterwards)."""
        self.model, self.device = model, device
        self.fp8 = bool(fp8)
        model.model.fp8_train = self.fp8
        self._calibrated = not self.fp8
        det = model.model.Detect
        hyp = dict(HYP)
        hyp['box'] *= 3 / det.nl                                    # train.py:158-165
        hyp['cls'] *= det.nc / 80
        hyp['obj'] *= (img_size / 640) ** 2 * 3 / det.nl
        self.obj_loss = ComputeObjectLoss(model, hyp, device)
        self.det_loss = ComputeDetectorLoss(device)
        # the reference's optimizer (train.py:88) in its single-kernel implementation: the default multi-tensor one re-reads the 7.6 M
        # parameters / moments in ~10 passes (2.5-3 ms of the step); same update rule, fp32
        self.gas, self.max_grad_norm = int(gas), max_grad_norm
        # gradient buckets in the order the gradients become final: the parameters only the full backward reaches first (all-reduced
        # while the keypoint-only backward of the warped pass runs), the shared trunk + keypoint / descriptor heads last
        from .training import grad_ready_groups, link_siblings
        groups = grad_ready_groups(model.model)
        link_siblings(model.model)
        self.reducer = GradAllReducer(None, group=group, groups=groups)
        kp = set(id(p) for p in groups[1][1])
        # YP_TRAIN_PAIR=0: the two forwards of a step as two native passes (two graphs, the schedule of round 1) instead of one 2B-sample pass
        self.pair = sw("YP_TRAIN_PAIR") != "0"
        # contributions a gradient receives per micro-batch: pair mode -- one backward plan reaches each parameter; two-graph mode -- the
        # trunk / keypoint-head parameters are reached by both passes' backward
        self.reducer.set_expected({p: (2 if (id(p) in kp and not self.pair) else 1) for p in self.reducer.params})
        # the reference's optimizer (train.py:88).  Default: optim.FlatAdam -- parameters / gradients / moments as four flat arrays, one
        # launch per step (torch's fused multi-tensor Adam: 6 launches, 0.38 ms for the 7.6 M parameters of YOLOPoint-s; its default
        # multi-tensor one re-reads them in ~10 passes, 2.5-3 ms).  YP_ADAM=torch: torch.optim.Adam(fused=True) over the same parameters.
        if sw("YP_ADAM") == "flat" and torch.device(device).type == "cuda":
            from .optim import FlatAdam
            self.opt = FlatAdam(self.reducer, params=[p for p in model.parameters() if p.requires_grad], lr=lr, all_params=list(model.parameters()))
        else:
            self.opt = torch.optim.Adam(model.parameters(), lr=lr, fused=torch.device(device).type == "cuda")
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.opt, lr_lambda=lr_lambda) if lr_lambda is not None else None
        self.sparse = dict(SPARSE)
        self.comm_events = None
        if torch.device(device).type == "cuda" and sw("YP_TRAIN_SIDE_STREAM") == "1":
            from . import _hip
            _hip.check(_hip.lib().yp_sampling_set_max_workgroups(int(sw("YP_SIDE_WGS"))))
        self._in_flight, self._max_in_flight = collections.deque(), int(sw("YP_STEPS_IN_FLIGHT"))
        # the loss / label stream: the stream the library TESTED to run beside the step's main stream (yp_stream_pick: streams that share a
        # hardware queue serialise -- a torch.cuda.Stream() from PyTorch's pool landed on the main stream's queue or not depending on how many
        # streams the process had handed out before: 7.4 or 9.4 ms per step).  It IS the plans' side lane (slot 0): the label kernels, the
        # heads of the forward and the InfoNCE chain are all background work for the main chain, and one queue for them measured faster than
        # two (7.45 vs 7.68 ms with a third queue; a PyTorch pool stream may share the main stream's queue)
        self._use_side = sw("YP_TRAIN_SIDE_STREAM") == "1" and torch.device(device).type == "cuda"
        self._side_streams = {}
        self.reducer.broadcast_parameters(model)

    def __call__(self, batch):
        with torch.cuda.device(self.device):
            return self._step(batch)

    def _side_for(self, main):
        if not self._use_side:
            return None
        key = main.cuda_stream
        s = self._side_streams.get(key)
        if s is None:
            import ctypes as C
            from . import _hip
            out = C.c_void_p()
            _hip.check(_hip.lib().yp_stream_pick(C.c_void_p(key), 0, C.byref(out)))
            s = self._side_streams[key] = torch.cuda.ExternalStream(out.value, device=self.device)
        return s

    def _step(self, batch):
        """batch: one micro-batch (gas == 1) or a sequence of `gas` micro-batches."""
        micro = list(batch) if isinstance(batch, (list, tuple)) else [batch]
        if len(micro) != self.gas:
            raise ValueError(f"TrainStep(gas={self.gas}) takes {self.gas} micro-batch(es) per optimizer step, got {len(micro)}")
        if not self._calibrated:
            from .models.common import invalidate_packed_weights
            # The two calibration passes only record tensor maxima: the BatchNorm running statistics / batch counters they update and the
            # random draws they consume (InfoNCE sampling) are put back, so an fp8 run starts from the same buffers and the same RNG
            # stream as the bf16 run it is compared with -- and as a resume from a checkpoint.
            buffers = [(b, b.detach().clone()) for b in self.model.buffers()]
            cpu_rng, dev_rng = torch.get_rng_state(), torch.cuda.get_rng_state(self.device)
            with self.reducer.no_sync():
                for _ in range(2):
                    self.loss_and_grads(micro[0])
                    invalidate_packed_weights()       # (the next forward turns the recorded maxima into scales, as after an optimizer step)
            with torch.no_grad():
                for b, saved in buffers:
                    b.copy_(saved)
            torch.set_rng_state(cpu_rng)
            torch.cuda.set_rng_state(dev_rng, self.device)
            self._calibrated = True
        total = None
        for i, mb in enumerate(micro):
            last = i == len(micro) - 1
            with (contextlib.nullcontext() if last else self.reducer.no_sync()):
                loss = self.loss_and_grads(mb, first_micro=(i == 0), scale=1.0 / len(micro))
            total = loss if total is None else total + loss
        if self.comm_events is not None:            # (bench: how long the compute stream waits for the collectives = exposed communication)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.reducer.finish()
            e1.record()
            self.comm_events.append((e0, e1))
        else:
            self.reducer.finish()                   # the compute stream waits for the collectives here, right before the optimizer reads
        if self.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.max_grad_norm)
        if self.fp8 and not getattr(self, "_fp8_grads_checked", False):
            # first fp8 optimizer step only (one host read-back): a 16-bit copy that training.TrainGraph._drop_unread_16bit_copies stopped
            # writing is NaN-filled, so a reader its analysis missed shows up HERE as a non-finite gradient instead of training on garbage
            self._fp8_grads_checked = True
            dropped = sum(getattr(gg, "n_twin_only", 0) for g_ in getattr(self.model.model, "_train_graphs", {}).values()
                          for gg in (g_ if isinstance(g_, (list, tuple)) else [g_]))
            if dropped and not bool(torch.isfinite(self.reducer.arena).all()):
                from ._hip import YpError
                raise YpError(f"fp8 training: non-finite gradients after the first step with {dropped} twin-only BatchNorm outputs -- something reads a "
                              f"16-bit copy the graph no longer writes (register it with TrainGraph.register_external_read, or set YP_FP8_TWIN_ONLY=0)")
        self.opt.step()
        # Nothing in the step makes the host wait for the device any more, so a loop that never reads a loss value would queue steps without
        # bound (and hold every step's side-stream buffers until the device catches up).  Back-pressure: at most YP_STEPS_IN_FLIGHT
        # optimizer steps are queued (default 2: the host stays a full step ahead, which is all the overlap there is to have).
        if self._max_in_flight > 0 and torch.device(self.device).type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._in_flight.append(ev)
            if len(self._in_flight) > self._max_in_flight:
                self._in_flight.popleft().synchronize()
        return total

    # ------------------------------------------------------------------------------------------------------------------------------
    # The loss stage without the framework: the native loss kernels read the heads where the forward plan left them and write the FINAL
    # head gradients where the backward plans read them; the label-only work is native kernels on the side stream; the loss sum of
    # train.py:232-241 is one tiny launch.  Same values as the autograd formulation below -- the gradients bit for bit
    # (tests/test_gpu_training.py) -- and no ATen / rocPRIM / hipBLASLt kernel in a steady-state step.  YP_NATIVE_STAGE=0: the autograd
    # formulation (also what YOLOPointv52, two-graph mode and CPU tensors use).
    # ------------------------------------------------------------------------------------------------------------------------------
    def loss_terms(self):
        """[total, detector, descriptor, object] of the last micro-batch as Python floats (one host read-back).  The device-count form of the
        native stage cannot raise when an image has no valid cell under the warp (the pool is empty; the reference would average over nothing:
        NaN) -- that step runs WITHOUT a descriptor term.  The native stage reports its InfoNCE row count beside the terms (`last_nce_rows`,
        0 = empty pool); the autograd formulation is recognised by a descriptor term of exactly zero."""
        t = getattr(self, "last_loss_terms", None)
        if t is None:
            return None
        import math
        v = [float(x) for x in t.detach().float().cpu().tolist()]
        rows = v[4] if len(v) >= 5 else None                 # native stage: the InfoNCE row count of the step (yp_loss_combine5's out5[4])
        v = v[:4]
        if rows is not None and not (math.isfinite(rows) and rows >= 0.0):
            rows = None                                      # (not a count: fall back to the descriptor-term test)
        self.last_nce_rows = None if rows is None else int(rows)
        if (rows == 0.0) if rows is not None else (len(v) >= 3 and v[2] == 0.0):
            import warnings
            warnings.warn("TrainStep: the last step ran without a descriptor (InfoNCE) term -- an image of the batch had no valid cell under its "
                          "homography (empty sampling pool); check the warps / valid masks of the batch")
        return v

    def _native_stage_ok(self, batch):
        if not self.pair or sw("YP_NATIVE_STAGE") == "0" or type(self.model.model).__name__ != "YOLOPoint":
            return False
        img = batch['image']
        ok = img.is_cuda and img.dim() == 4 and img.shape[-1] % 8 == 0 and img.shape[-2] % 8 == 0 and tuple(batch['warped_image'].shape) == tuple(img.shape)
        # the stage consumes the device-side sampling (utils.loss_functions._prepare_native with the sorted pair index): the same predicate
        # as infonce_prepare's, or the step would trip over `assert sync` / a 4-tuple in the middle of a step with the graph marked busy
        ok = ok and (img.shape[-2] // 8) * (img.shape[-1] // 8) < 36864 and sw("YP_NATIVE_PREPARE") != "0"
        for k in ('labels_2D', 'warped_labels', 'valid_mask', 'warped_valid_mask'):
            t = batch[k]
            ok = ok and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == img.shape[0] * img.shape[-2] * img.shape[-1]
        return bool(ok)

    def _det_strides(self):
        # Detect.stride lives on the device: reading it per step would make the host wait for the forward it has just launched
        st = getattr(self, "_det_strides_host", None)
        if st is None:
            st = self._det_strides_host = [int(v) for v in self.model.model.Detect.stride.tolist()]
        return st

    def _loss_and_grads_native(self, batch, first_micro, scale):
        import numpy as np
        from types import SimpleNamespace
        from . import _hip
        from .training import run_native_backward_pair, SEEDED
        lib, check, sp = _hip.lib(), _hip.check, _hip.stream_ptr
        m, dev = self.model, self.device
        net = m.model
        self.reducer.bind_grads(zero=first_micro)
        img, img_w = batch['image'].contiguous().float(), batch['warped_image'].contiguous().float()
        B, H, W = img.shape[0], img.shape[-2], img.shape[-1]
        Hc, Wc = H // 8, W // 8
        main = torch.cuda.current_stream(dev)
        side = self._side_for(main) or main
        fork = main.record_event() if side is not main else None
        g = net._train_graph(img, pair=True, fp8=bool(getattr(net, "fp8_train", False)))
        g.busy = True
        # Enqueue ORDER of the side lane (YP_LABELS_ORDER).  The label-only kernels share the side stream with the forward plan's head ops.
        #   "after" (rounds 3-4): all of them enqueued behind the forward launch -- since the heads moved to this stream they queue behind
        #           the heads, run AFTER the forward, and the main lane waits for all of them in front of the object loss (lane-annotated
        #           trace, tools/profile_collect.py: ONE pause of 1.12 ms per -s step with only the side lane busy);
        #   "first": all of them in front of the forward launch -- the heads then start ~1.1 ms late and the main lane waits for THEM;
        #   "split" (default): target assignment + cell masks (what the object / detector losses read: ~60 us) in front of the forward
        #           launch, the InfoNCE sampling and its two CSR sorts (~1 ms, read by the InfoNCE chain on this same stream only) behind
        #           the heads; the main lane waits for the small part only.
        order = sw("YP_LABELS_ORDER") if side is not main else "after"
        if order == "split" and int(sw("YP_LOSS_LANES")) < 2:
            order = "after"                       # (the InfoNCE chain runs on the main stream there: it needs the sampling joined)
        if order == "after":
            g.forward(img, img_w, export=False)
        D = g.desc_channels
        stg = getattr(g, "_stage", None)
        if stg is None:
            wsb = lib.yp_cell_mask_workspace_bytes(B, H, W)
            sv = g.semi_v.buf.t[..., g.semi_v.coff:g.semi_v.coff + 65].permute(0, 3, 1, 2)
            gd = getattr(g, "seed_desc_buf", None)
            if not (sv.dtype == torch.float32 and g.seed_semi.dtype == torch.float32 and gd is not None and gd.C == D and g.desc_v.coff == 0
                    and g.desc_v.buf.C == D and g.desc_v.buf.t.dtype == torch.float32 and D % 64 == 0 and D <= 256):
                raise _hip.YpError("native loss stage: unexpected head buffer layout (set YP_NATIVE_STAGE=0)")
            stg = g._stage = SimpleNamespace(
                ws_bytes=wsb, ws_side=torch.empty(wsb, dtype=torch.uint8, device=dev), ws_main=torch.empty(wsb, dtype=torch.uint8, device=dev),
                mask=torch.empty((2, B, Hc, Wc), dtype=torch.float32, device=dev), scal=torch.zeros(16, dtype=torch.float32, device=dev),
                semi_ptr=sv.data_ptr(), zs=(C.c_int64 * 4)(*sv.stride()), dsemi_ptr=g.seed_semi.data_ptr(), ds=(C.c_int64 * 4)(*g.seed_semi.stride()),
                desc_ptr=g.desc_v.buf.t.data_ptr(), gdesc_ptr=gd.t.data_ptr())
        det = net.Detect
        shapes = [(B, det.na, H // st, W // st, det.no) for st in self._det_strides()]
        scal = stg.scal.data_ptr()           # floats: [0:3] object-loss sums, [4:6] detector losses, [6:8] mask sums, [12] InfoNCE gradient scale

        def small_labels():
            tgt_ = self.obj_loss.assign(shapes, batch['box_labels'])
            for j, key in enumerate(('valid_mask', 'warped_valid_mask')):
                check(lib.yp_cell_mask(batch[key].data_ptr(), B, H, W, stg.mask[j].data_ptr(), scal + 4 * (6 + j), stg.ws_side.data_ptr(), stg.ws_bytes, sp()))
            return tgt_

        def nce_labels():
            return infonce_prepare(batch['warped_valid_mask'], batch['inv_homographies'], (B, D, Hc, Wc), True, self.sparse['num_samples_per_image'],
                                   self.sparse['num_masked_non_matches_per_match'], 8, dev, pair_index=True,
                                   sync=sw("YP_PREPARE_SYNC") == "1" or sw("YP_NATIVE_PREPARE") == "0")
        if side is not main:
            side.wait_event(fork)
            with torch.cuda.stream(side):
                tgt_e = small_labels()
                _record_stream(tgt_e, main)
                small_done = side.record_event()
                nce_e = nce_labels() if order != "split" else None
                labels_done = side.record_event()
            if order != "after":
                g.forward(img, img_w, export=False)
            if order == "split":
                with torch.cuda.stream(side):
                    nce_e = nce_labels()          # behind the forward's head ops in the side stream's order
                main.wait_event(small_done)
            else:
                _record_stream(nce_e, main)
                main.wait_event(labels_done)
            early = (tgt_e, nce_e)
        else:
            early = (small_labels(), nce_labels())
        tgt, nce = early
        f32 = np.float32
        # ---- object loss (reference utils/loss_functions.py:90-176): value into scal[0:3], UNSCALED gradient straight into the backward
        # plan's Detect seeds; its upstream factor lambda_obj * scale is applied by the plan's first op (TrainGraph.head_scale)
        want = float(f32(f32(scale) * f32(LAMBDA_OBJ))) if scale != 1.0 else float(f32(LAMBDA_OBJ))
        g.set_head_scale(want)
        # The object loss and the detector loss read other heads than InfoNCE and write other seeds (YP_LOSS_LANES below).
        def object_loss():
            check(lib.yp_fill_zero(scal, 16, sp()))
            ol, hyp = self.obj_loss, self.obj_loss.hyp
            cap = tgt["cap"] if tgt["nt"] else 0
            cap_alloc = max(cap, 1)
            for i, xo in enumerate(g.xs):
                no = xo.shape[-1]
                cells = (xo.numel() // xo.shape[0]) * B // no
                iou = torch.empty((cap_alloc,), dtype=torch.float32, device=dev)
                own = torch.empty((cells,), dtype=torch.int32, device=dev)
                check(lib.yp_objloss_level_dev(xo.data_ptr(), cells, no, ol.nc, tgt["cell"].data_ptr() + 4 * i * tgt["cap"], tgt["box"].data_ptr() + 16 * i * tgt["cap"],
                                               tgt["anchor"].data_ptr() + 8 * i * tgt["cap"], tgt["cls"].data_ptr() + 4 * i * tgt["cap"], cap,
                                               tgt["count"].data_ptr() + 4 * i, float(ol.cp), float(ol.cn), float(hyp['cls_pw']), float(hyp['obj_pw']),
                                               float(hyp['box']), float(hyp['obj']) * float(ol.balance[i]), float(hyp['cls']), iou.data_ptr(), own.data_ptr(),
                                               g.g_xs[i].data_ptr(), scal, sp()))
        # ---- detector loss of both passes (utils/loss_functions.py:600-619 on labels2Dto3D / getMasks of the 2-D maps): final gradients
        # into the semi seed
        def detector_losses():
            for j, key in enumerate(('labels_2D', 'warped_labels')):
                check(lib.yp_detloss2d(stg.semi_ptr + 4 * j * B * stg.zs[0], stg.zs, batch[key].data_ptr(), stg.mask[j].data_ptr(), scal + 4 * (6 + j), float(f32(scale)),
                                       B, H, W, stg.dsemi_ptr + 4 * j * B * stg.ds[0], stg.ds, scal + 4 * (4 + j), stg.ws_main.data_ptr(), stg.ws_bytes, sp()))
        lanes = int(sw("YP_LOSS_LANES")) if side is not main else 0
        tau = 0.07
        g_desc = f32(f32(scale) * f32(LAMBDA_DESC)) if scale != 1.0 else f32(LAMBDA_DESC)

        # ---- InfoNCE (utils/loss_functions.py:484-597): one lookup over both passes' descriptor maps, loss rows + anchor-side gradient in
        # one gather pass, the loss sum, the match-side gradient, the scatter into the descriptor seed
        def infonce_chain(stream, small_done):
            # (beside the backward plan the gathers get 3 workgroups per CU: uncapped they take every slot and the plan's kernels queue behind
            # them -- -s 7.85 -> 7.55 ms, -l 35.8 -> 35.4 at 768; 1024: 7.67, 512: 7.66, 256: 9.06)
            # second-generation gathers (8 load instructions in flight per wave, csrc/losses.hip): they hold their rate with far fewer
            # workgroups, and the fewer they occupy the less the backward plan beside them is slowed -- until the chain itself becomes the
            # critical path.  -s (D = 128): 256 workgroups 7.32 ms, 128 / 384 / 512 / 768: 7.36 / 7.38 / 7.42 / 7.55; -l (D = 256, 16
            # samples): 128 workgroups 34.6-34.8 ms, 256 / 384 / 768: 34.9 / 35.0 / 35.5, 96 / 64 / 32: 35.3 / 38.5 / 49.6
            # (16-bit rows, D = 256: 96 / 128 / 192 workgroups 29.94 / 30.35 / 29.96 ms per -l fp8 step, same box)
            nce_wgs = int(sw("YP_NCE_WGS") or ("256" if D <= 128 else ("96" if g.code == _hip.YP_BF16 and sw("YP_NCE_ROWS") == "bf16" else "128"))) if lanes >= 2 else 0
            out4_ = torch.empty((8,), dtype=torch.float32, device=dev)        # [total, detector, descriptor, object, InfoNCE row count, -, -, -]: yp_loss_combine5 writes all five on both count paths
            if isinstance(nce, dict):
                # counts on the device (infonce_prepare(sync=False)): arrays at their capacity, the kernels read points-per-image / matched rows
                # from the sampling's meta words -- no host synchronisation anywhere in the step
                meta = nce["meta"].data_ptr()
                p_dev, n_dev = meta, meta + 4
                n, pool, E = nce["n_cap"], nce["pool_cap"], nce["E"]
                uab, idx, order, offsets, s_order, s_offsets = (nce[k] for k in ("uab", "idx", "order", "offsets", "s_order", "s_offsets"))
                desc_scale = 0.0
            else:
                ua, _, _, (idx, order, offsets), (uab, s_order, s_offsets) = nce
                p_dev = n_dev = None
                pool, E = ua.shape[1], idx.shape[1]
                n = B * pool
                desc_scale = float(f32(g_desc * f32(1.0 / (tau * n))))
            dab = torch.empty((2 * n, D), dtype=torch.float32, device=dev)
            grad = torch.empty((2 * n, D), dtype=torch.float32, device=dev)
            w = torch.empty((n, E), dtype=torch.float32, device=dev)
            rows, lse = torch.empty((n,), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev)
            check(lib.yp_points_sample_fwd(stg.desc_ptr, 2 * B, Hc, Wc, D, uab.data_ptr(), pool, dab.data_ptr(), p_dev, sp()))
            # bf16 graphs (incl. fp8 mode): the gathers read a 16-bit copy of the sampled-descriptor table -- half the gathered bytes (the rows
            # of these two kernels are a fifth of the -l step's HBM / fabric traffic); fp32 / f16 graphs keep the fp32 rows.  YP_NCE_ROWS=fp32: off
            rows16 = None
            if g.code == _hip.YP_BF16 and D in (64, 128, 256) and sw("YP_NCE_ROWS") == "bf16":
                rows16 = torch.empty((2 * n, D), dtype=torch.bfloat16, device=dev)
                check(lib.yp_infonce_rows16(dab.data_ptr(), 2 * n * D, rows16.data_ptr(), sp()))
                check(lib.yp_infonce_fwd_grad_h(dab.data_ptr(), rows16.data_ptr(), idx.data_ptr(), n, E, D, 1.0 / tau, w.data_ptr(),
                                                rows.data_ptr(), lse.data_ptr(), grad.data_ptr(), n_dev, nce_wgs, sp()))
            else:
                check(lib.yp_infonce_fwd_grad(dab.data_ptr(), None if n_dev else dab.data_ptr() + 4 * n * D, idx.data_ptr(), n, E, D, 1.0 / tau, w.data_ptr(),
                                              rows.data_ptr(), lse.data_ptr(), grad.data_ptr(), n_dev, nce_wgs, sp()))
            if small_done is not None:
                stream.wait_event(small_done)
            check(lib.yp_loss_combine5(scal + 16, 2, rows.data_ptr(), n, scal, LAMBDA_DESC, LAMBDA_OBJ, float(scale), desc_scale, out4_.data_ptr(), scal + 48, n_dev,
                                      float(g_desc), tau, sp()))
            if rows16 is not None:
                check(lib.yp_infonce_bwd_db_h(rows16.data_ptr(), order.data_ptr(), offsets.data_ptr(), w.data_ptr(), n, E, D, scal + 48,
                                              grad.data_ptr() if n_dev else grad.data_ptr() + 4 * n * D, n_dev, nce_wgs, sp()))
            else:
                check(lib.yp_infonce_bwd_db(dab.data_ptr(), order.data_ptr(), offsets.data_ptr(), w.data_ptr(), lse.data_ptr(), n, E, D, scal + 48,
                                            grad.data_ptr() if n_dev else grad.data_ptr() + 4 * n * D, n_dev, nce_wgs, sp()))
            check(lib.yp_points_sample_bwd_sorted(grad.data_ptr(), 2 * B, Hc, Wc, D, uab.data_ptr(), pool, s_order.data_ptr(), s_offsets.data_ptr(), scal + 48, n,
                                                  stg.gdesc_ptr, n_dev, sp()))
            return out4_

        join = None
        if lanes >= 2:
            # The whole InfoNCE chain (latency-bound gathers, 0.8 ms at -s / 2.5 ms at -l) on the side stream, beside the YOLO-branch backward
            # plan, which only needs the Detect seeds of the object loss; the trunk plan waits for it.
            fwd_done = main.record_event()
            object_loss()                           # (the only loss the YOLO-branch plan waits for)
            obj_done = main.record_event()
            side.wait_event(fwd_done)
            with torch.cuda.stream(side):
                detector_losses()
                det_done = side.record_event()
                out4 = infonce_chain(side, obj_done)
                out4.record_stream(main)
            nce_done = side.record_event()
            join = lambda: main.wait_event(nce_done)
        elif lanes == 1:
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                object_loss()
                detector_losses()
            out4 = infonce_chain(main, side.record_event())
        else:
            object_loss()
            detector_losses()
            out4 = infonce_chain(main, None)
        # ---- backward: YOLO-branch plan -> its buckets go out (the InfoNCE lane joins) -> trunk plan over both passes
        self.reducer.begin()
        run_native_backward_pair(g, SEEDED, SEEDED, [SEEDED] * len(g.xs), notify=self.reducer.notify, join=join)
        self.last_loss_terms = out4              # [total, detector, descriptor, object] (device)
        return out4[0]

    def loss_and_grads(self, batch, prepare=True, first_micro=True, scale=1.0):
        """loss = (det + det_warp) + lambda_desc * infonce + lambda_obj * obj and its backward (reference train.py:208-245).
        With `prepare`, the label-only parts of the losses (YOLO target assignment: a device kernel; InfoNCE sampling: device kernels, counts stay on the device)
        run right after both forward passes have been launched.  The backward is driven explicitly: the loss kernels are
        differentiated down to the network's head outputs (torch.autograd.grad), then the FULL native backward of the image pass
        runs, the gradient buckets it completes are handed to the all-reduce, and the keypoint-only backward of the warped pass runs
        while those collectives are in flight (dp.GradAllReducer).
        (Measured and dropped: a two-stage backward that launches the warped pass's native backward before the
        object-loss backward is differentiated -- the step is device-bound, the extra autograd entry points cost more than the
        overlap wins: 40-48 ms vs 38 ms per step.)"""
        from .training import run_native_backward, run_native_backward_pair
        m, dev = self.model, self.device
        if prepare and self._native_stage_ok(batch):
            try:
                return self._loss_and_grads_native(batch, first_micro, scale)
            except BaseException:
                for g in getattr(m.model, "_train_graphs", {}).values():      # an exception inside the stage must not leak a busy graph
                    for gg in (g if isinstance(g, (list, tuple)) else [g]):
                        if hasattr(gg, "busy"):
                            gg.busy = False
                raise
        self.reducer.bind_grads(zero=first_micro)   # (instead of optimizer.zero_grad: gradients accumulate straight into the all-reduce buckets)
        img = batch['image']
        B, S = img.shape[0], img.shape[-1]
        # The label-only parts of the losses (YOLO target assignment, InfoNCE sampling, the 65-channel
        # keypoint labels and cell masks: ~100 small launches that depend on the batch only) run on a side stream that forks from the main
        # stream at a point BEFORE the forward: they execute beside the forward pass instead of between it and the losses, and the host
        # synchronisation waits for the side stream only.  YP_TRAIN_SIDE_STREAM=0: everything on the main stream, after the forward launch.
        main = torch.cuda.current_stream(dev)
        side = (self._side_for(main) if prepare else None) or main
        fork = main.record_event() if side is not main else None      # (the batch tensors were produced on the main stream before this point)

        def label_work():
            det = m.model.Detect
            shapes = [(B, det.na, img.shape[-2] // st, S // st, det.no) for st in self._det_strides()]
            tgt_ = self.obj_loss.assign(shapes, batch['box_labels']) if prepare else None
            dch = getattr(m.model, "_desc_channels", None) or m.model.ConvDesc.out_channels
            nce_ = infonce_prepare(batch['warped_valid_mask'], batch['inv_homographies'], (B, dch, img.shape[-2] // 8, S // 8), True,
                                   self.sparse['num_samples_per_image'], self.sparse['num_masked_non_matches_per_match'], 8, dev, pair_index=self.pair) if prepare else None
            if self.pair:       # both passes' labels / masks as one 2B batch (image pass first)
                lab_ = (labels2Dto3D(torch.cat((batch['labels_2D'], batch['warped_labels']))),
                        getMasks(torch.cat((batch['valid_mask'], batch['warped_valid_mask'])), dev), None, None)
            else:
                lab_ = (labels2Dto3D(batch['labels_2D']), getMasks(batch['valid_mask'], dev),
                        labels2Dto3D(batch['warped_labels']), getMasks(batch['warped_valid_mask'], dev))
            return tgt_, nce_, lab_
        if self.pair:
            outs, outs_w, raw, graph = m.model.forward_pair(img, batch['warped_image'])
        else:
            outs, raw, graph = m.model.forward_with_graph(img)
            outs_w, raw_w, graph_w = m.model.forward_with_graph(batch['warped_image'])
        # (enqueued after the forward launch so that the host never waits before the device has the forward to work on; on the device the
        # side stream depends on the fork event only)
        if side is not main:
            side.wait_event(fork)
            with torch.cuda.stream(side):
                early = label_work()
                _record_stream(early, main)
            main.wait_stream(side)
        else:
            early = label_work()
        tgt, nce, (lab, msk, lab_w, msk_w) = early
        l_obj = self.obj_loss(outs['objects'], batch['box_labels'], prepared=tgt)[0]
        if self.pair:
            # both passes' heads are halves of one tensor: the detector loss of both in one call (sum of the two per-pass losses), the
            # descriptors sampled by one launch -- each head receives ONE gradient tensor in its own memory layout
            l_dets = self.det_loss(outs['semi_pair'], lab, msk, groups=2)
            l_desc = infonce(outs['desc'], outs_w['desc'], batch['warped_valid_mask'], batch['inv_homographies'], device=dev, prepared=nce,
                             descriptors_pair=outs['desc_pair'], **self.sparse)
        else:
            l_dets = self.det_loss(outs['semi'], lab, msk) + self.det_loss(outs_w['semi'], lab_w, msk_w)
            l_desc = infonce(outs['desc'], outs_w['desc'], batch['warped_valid_mask'], batch['inv_homographies'], device=dev, prepared=nce, **self.sparse)
        loss = l_dets + LAMBDA_DESC * l_desc + LAMBDA_OBJ * l_obj
        if scale != 1.0:
            loss = loss * scale                     # accelerator.backward divides by the accumulation steps
        if self.pair:
            g = torch.autograd.grad(loss, list(raw), allow_unused=True)
            self.reducer.begin()
            # YOLO-branch plan over the image pass -> its (detector-group) buckets go out -> trunk plan over both passes
            run_native_backward_pair(graph, g[0], g[1], list(g[2:]), notify=self.reducer.notify)
            return loss.detach()
        heads = list(raw) + list(raw_w[:2])
        g = torch.autograd.grad(loss, heads, allow_unused=True)
        self.reducer.begin()
        self.reducer.notify(run_native_backward(graph, g[0], g[1], list(g[2:len(raw)])))           # image pass: the whole network
        self.reducer.notify(run_native_backward(graph_w, g[len(raw)], g[len(raw) + 1], [None] * (len(raw_w) - 2)))   # warped pass: keypoint sub-graph
        return loss.detach()
