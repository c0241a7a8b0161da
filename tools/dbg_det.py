import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import make_model
from oracle import net_oracle, postproc_oracle as po
from yolopoint_amd.utils.general_yolo import non_max_suppression
for dtype in ("f32","f16"):
    m, sd = make_model("s", 77, dtype=dtype)
    for i in range(3): sd[f"model.Detect.m.{i}.weight"] = sd[f"model.Detect.m.{i}.weight"]*0.1
    m.load_state_dict(sd, strict=True); m=m.cuda()
    x = net_oracle.synth_image(2,3,256,256,5)
    with torch.no_grad():
        ref = net_oracle.yolopoint_forward(sd, x, "s"); got = m(x.cuda())
    p, q = ref["objects"][0], got["objects"][0].cpu()
    print(dtype, "box abs err max", float((p[...,:4]-q[...,:4]).abs().max()), "obj/cls abs err max", float((p[...,4:]-q[...,4:]).abs().max()), "mean", float((p[...,4:]-q[...,4:]).abs().mean()))
    # NMS of the SAME tensor (oracle pred) through both implementations
    a = po.non_max_suppression(p.numpy(), 0.7, 0.45, agnostic=False, multi_label=False, max_det=2000)
    b = non_max_suppression(p.cuda(), 0.7, 0.45, labels=[], multi_label=False, agnostic=False, max_det=2000)
    print("   same-input NMS equal:", all(np.array_equal(u, v.cpu().numpy()) for u, v in zip(a, b)), [len(u) for u in a])
    c = non_max_suppression(got["objects"][0], 0.7, 0.45, labels=[], multi_label=False, agnostic=False, max_det=2000)
    print("   counts oracle-pred vs hip-pred:", [len(u) for u in a], [len(v) for v in c])
    # candidates above threshold
    conf_p = (p[...,5:]*p[...,4:5]).max(-1)[0]; conf_q=(q[...,5:]*q[...,4:5]).max(-1)[0]
    print("   candidates >0.7:", int((conf_p>0.7).sum()), int((conf_q>0.7).sum()), "both", int(((conf_p>0.7)&(conf_q>0.7)).sum()))
    from oracle import eval_oracle as eo
    for u, v in zip(a, c):
        v = v.cpu().numpy()
        iou = eo.box_iou(u[:, :4], v[:, :4])
        ok = (iou >= 0.5) & (u[:, 5:6] == v[None, :, 5]) & (np.abs(u[:, 4:5] - v[None, :, 4]) <= 0.05)
        print("   twins", int(ok.any(1).sum()), len(u), " exact rows equal:", int((np.abs(u[:len(v)] - v[:len(u)]).max(1) < 1e-3).sum()) if len(u)==len(v) else -1)
        bad = np.where(~ok.any(1))[0][:3]
        for i in bad:
            j = iou[i].argmax(); print("      no twin:", u[i], " best hip:", v[j], "iou", iou[i, j])
