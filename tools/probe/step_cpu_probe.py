"""Host-side time of each phase of a training step WITHOUT synchronising in between (how long the host spends issuing
each phase), then the final wait for the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch, LAMBDA_DESC, LAMBDA_OBJ
from yolopoint_amd.utils.loss_functions import infonce, infonce_prepare
from yolopoint_amd.utils.utils import labels2Dto3D, getMasks
dev = torch.device("cuda:0")
m, _ = make_model("s", 1, dtype="bf16"); m = m.to(dev).train()
step = TrainStep(m, dev, img_size=640)
batch = synthetic_batch(8, 640, dev, 1234)
for _ in range(3): step(batch)
torch.cuda.synchronize()
acc = {}
def lap(name, t=[0.0]):
    now = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (now - t[0]); t[0] = now
N = 10
for _ in range(N):
    lap("_")
    step.opt.zero_grad(set_to_none=True); lap("zero_grad")
    img = batch['image']; B, S = img.shape[0], img.shape[-1]
    det = m.model.Detect
    shapes = [(B, det.na, S // int(st), S // int(st), det.no) for st in det.stride]
    tgt = step.obj_loss.build_targets(shapes, batch['box_labels']); lap("build_targets")
    nce = infonce_prepare(batch['warped_valid_mask'], batch['inv_homographies'], (B, m.model.ConvDesc.out_channels, S // 8, S // 8), True,
                          step.sparse['num_samples_per_image'], step.sparse['num_masked_non_matches_per_match'], 8, dev); lap("infonce_prepare")
    o = m(img); lap("forward 1")
    ow = m(batch['warped_image']); lap("forward 2")
    lo = step.obj_loss(o['objects'], batch['box_labels'], prepared=tgt)[0]; lap("object loss")
    ld = step.det_loss(o['semi'], labels2Dto3D(batch['labels_2D']), getMasks(batch['valid_mask'], dev)) + \
         step.det_loss(ow['semi'], labels2Dto3D(batch['warped_labels']), getMasks(batch['warped_valid_mask'], dev)); lap("detector losses")
    ln = infonce(o['desc'], ow['desc'], None, None, device=dev, prepared=nce); lap("infonce")
    loss = ld + LAMBDA_DESC * ln + LAMBDA_OBJ * lo
    loss.backward(); lap("backward")
    step.reducer.all_reduce(); lap("all_reduce")
    step.opt.step(); lap("adam")
    torch.cuda.synchronize(); lap("final device wait")
tot = sum(v for k, v in acc.items() if k != "_")
for k, v in acc.items():
    if k != "_": print(f"  {k:22s} {v / N * 1e3:7.2f} ms")
print(f"  {'total':22s} {tot / N * 1e3:7.2f} ms")
