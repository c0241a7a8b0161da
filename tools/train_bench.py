#!/usr/bin/env python3
"""Time the full training step (tools-level probe; bench.py --mode train uses the same TrainStep)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--version", default="s"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=640)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
fp8 = a.dtype == "fp8"          # configs[4]: 8-bit Conv operands on top of the bf16 path
m, _ = make_model(a.version, 1, dtype="bf16" if fp8 else a.dtype)
m = m.to(dev).train()
step = TrainStep(m, dev, img_size=a.size, fp8=fp8)
batch = synthetic_batch(a.batch, a.size, dev, 1234)
t0 = time.perf_counter(); l = step(batch); torch.cuda.synchronize(); print(f"first step (plan build + autotune): {time.perf_counter()-t0:.1f}s loss={float(l):.4f}")
l = step(batch); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    l = step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"YOLOPoint-{a.version} train step B={a.batch} {a.size}x{a.size} {a.dtype}: {dt*1e3:.1f} ms/step = {a.batch/dt:.1f} samples/s ({2*a.batch/dt:.1f} images/s), loss={float(l):.4f}")
g = next(iter(m.model._train_graphs.values()))[0]
# (the first graph of the pool is the pair graph of TrainStep: forward over 2B samples, YOLO-branch backward plan, trunk backward plan)
PLANS = (("fwd", g.fwd_plan), ("bwd_yolo", g.bwd_plan), ("bwd_trunk", g.bwd_kp_plan))
for name, plan in PLANS:
    ms = plan.profile()
    print(f"  {name} plan: {len(ms)} launches, {sum(ms):.2f} ms eager")
    top = sorted(zip(ms, [r.name for r in plan.records]), reverse=True)[:8]
    print("    slowest:", ", ".join(f"{n}={t*1e3:.0f}us" for t, n in top))

# ---- host-side breakdown of one step (synchronising between phases)
import contextlib
from yolopoint_amd.utils.loss_functions import infonce
from yolopoint_amd.utils.utils import labels2Dto3D, getMasks
from yolopoint_amd.engine import SPARSE, LAMBDA_DESC, LAMBDA_OBJ
def tick(label, t=[time.perf_counter()]):
    torch.cuda.synchronize(); now = time.perf_counter(); print(f"    {label:28s} {(now - t[0])*1e3:7.1f} ms"); t[0] = now
print("  phase breakdown (synchronised):")
step.reducer.bind_grads(zero=True); tick("zero_grad (bind_grads)")
o = m(batch['image']); tick("forward 1")
ow = m(batch['warped_image']); tick("forward 2")
lo = step.obj_loss(o['objects'], batch['box_labels'])[0]; tick("object loss")
ld = step.det_loss(o['semi'], labels2Dto3D(batch['labels_2D']), getMasks(batch['valid_mask'], dev)) + step.det_loss(ow['semi'], labels2Dto3D(batch['warped_labels']), getMasks(batch['warped_valid_mask'], dev)); tick("detector losses")
ln = infonce(o['desc'], ow['desc'], batch['warped_valid_mask'], batch['inv_homographies'], device=dev, **SPARSE); tick("infonce")
loss = ld + LAMBDA_DESC * ln + LAMBDA_OBJ * lo
loss.backward(); tick("backward")
step.opt.step(); tick("adam")
gg = next(iter(m.model._train_graphs.values()))[0]
t0 = time.perf_counter(); gg.fwd_plan.refresh(); gg.bwd_plan.refresh(); torch.cuda.synchronize(); print(f"    weight refresh (1 graph)      {(time.perf_counter()-t0)*1e3:7.1f} ms")
t0 = time.perf_counter(); [fn() for fn in gg.collect]; torch.cuda.synchronize(); print(f"    grad collect (1 graph)        {(time.perf_counter()-t0)*1e3:7.1f} ms")
t0 = time.perf_counter(); gg.bwd_plan.run(); torch.cuda.synchronize(); print(f"    bwd plan run (1 graph)        {(time.perf_counter()-t0)*1e3:7.1f} ms")

# ---- aggregated per-kind table of both plans
import collections
for f_ in ("train_fwd_ops.txt", "train_bwd_ops.txt"):
    open(os.path.join(ROOT, "gpurun_out", f_), "w").close()
for name, plan in PLANS:
    ms = plan.profile()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for t, r in zip(ms, plan.records):
        key = (r.kind, r.name.split(".")[-1] if r.name else "")
        agg[key][0] += 1; agg[key][1] += t
    print(f"  {name} plan by kind:")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k[0]:8s} {k[1]:24s} n={n:4d} {t:8.3f} ms")
    with open(os.path.join(ROOT, "gpurun_out", "train_fwd_ops.txt" if name == "fwd" else "train_bwd_ops.txt"), "a") as f:
        f.write(f"# {name} plan (pair graph), {len(ms)} launches, {sum(ms):.3f} ms eager\n")
        for t, r in sorted(zip(ms, plan.records), key=lambda x: -x[0]):
            f.write(f"{t*1e3:9.1f} us  {r.kind:8s} {r.name:50s} M={r.M} N={r.N} K={r.K} flops={r.flops}\n")

# ---- loss-graph-only timings on detached head outputs
def det(o):
    return {'semi': o['semi'].detach().requires_grad_(), 'desc': o['desc'].detach().requires_grad_(), 'objects': [t.detach().requires_grad_() for t in o['objects']]}
od, owd = det(o), det(ow)
print("  loss graphs alone (detached heads, synchronised):")
tick("-")
for rep in range(2):
    lo = step.obj_loss(od['objects'], batch['box_labels'])[0]; tick("object loss fwd")
    lo.backward(); tick("object loss bwd")
    ln = infonce(od['desc'], owd['desc'], batch['warped_valid_mask'], batch['inv_homographies'], device=dev, **SPARSE); tick("infonce fwd")
    ln.backward(); tick("infonce bwd")
    ld = step.det_loss(od['semi'], labels2Dto3D(batch['labels_2D']), getMasks(batch['valid_mask'], dev)); tick("detector loss fwd")
    ld.backward(); tick("detector loss bwd")

if os.environ.get("YP_PROFILE_LOSS"):
    from torch.profiler import profile, ProfilerActivity
    for name, fn in (("infonce", lambda: infonce(od['desc'], owd['desc'], batch['warped_valid_mask'], batch['inv_homographies'], device=dev, **SPARSE)),
                     ("object", lambda: step.obj_loss(od['objects'], batch['box_labels'])[0])):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            l = fn(); l.backward(); torch.cuda.synchronize()
        print(f"==== {name}")
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=8, max_name_column_width=60))
