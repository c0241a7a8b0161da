#!/usr/bin/env python3
"""bench.py -- images/sec of the YOLOPoint hot path on MI355X.

`python bench.py [--gpus N --steps K --warmup W]` prints ONE JSON line.

Top level (BASELINE.json configs[1], the configuration `metric` is quoted on that fits one GPU): YOLOPoint-s inference, batch 8 per GPU,
640x640, fp16 compute (fp32 accumulate, fp32 head outputs), BN folded, seeded synthetic weights, synthetic images resident in HBM when
the timed region starts.  One step = one forward of the batch through the native plan (fused stem reading the NCHW fp32 batch, 47
convolution launches, SPPF pooling, descriptor L2 norm, Detect decode to [8, 25200, 85]) replayed on the plan's two lanes.  N > 1: inference
shards by independent images -- one replica per rank, no data-path collective, `scaling: weak`.

Launch forms.  `python bench.py --gpus N` (no launcher around it) starts the N ranks itself: it re-executes as `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`, one rank per GPU over RCCL (the reference's
`accelerate launch src/train.py`, README.md:76); under a launcher (WORLD_SIZE set -- the driver's form) it takes the launcher's ranks and exits with
code 2 when `--gpus` disagrees with WORLD_SIZE or the box has fewer than N GPUs.  Rank 0 prints the ONE line (`n_gpus`: N).  `--backend gloo --dry`:
launcher + process group + timed-region protocol with a stub step, no GPU (tests/test_bench_launcher.py); `--rehearsal`: everything for real with
the ranks sharing the visible GPU over gloo (control-flow check; tests/test_gpu_dp_tuning.py).
The numbers of every sub-record also appear as short keys inside `config` and `roofline` (bs1_ms, train_ms, train_bs64_ms, l_fp8_ms, l_bf16_ms,
fp8_over_bf16, *_frac, backbone_frac, *_grad_l2) and once more in `summary`.

Sub-records of the same line (each measured in this process, after the top-level timing):
  parity        head outputs of the timed plan against the oracle's fp32 CPU forward on the same weights and input (the oracle forward
                is what `cpu_baseline` times anyway): relative errors of semi / desc / raw Detect levels / decoded rows, argmax agreement
  roofline      the convolution kernels of the timed plan: algorithmic conv FLOP per forward / the sum of the conv
                launches' durations, HIP events on the launch stream; `backbone` = the same for Conv1..SPPooling (F_bb)
  cpu_baseline  the oracle's PyTorch-CPU fp32 forward timed on this host's cores (rank 0, N = 1), >= 3 warm-ups
  train         BASELINE configs[2]: the reference's optimizer step (train.py:189-259: two train-mode forwards, detector + object + InfoNCE
                losses, native backward, bucketed gradient all-reduce overlapped with the trunk backward plan, one-launch Adam), 8 samples per
                GPU, 640x640, bf16 -- DATA PARALLEL over all N ranks (weak scaling: 8 samples per GPU); reports the bucket plan, payload
                and the time the compute stream waits for the collectives (exposed communication)
  train_bs64    (N = 1) the metric's "train bs=64" on one GPU: ONE batch of 64 samples per optimizer step (train.py:38-43 with
                train_batch_size 64: gas = 1); `gas8_ms_per_step` = the same nominal batch as 8 micro-batches of 8 (train_batch_size 8)
  (train / train_bs64 / train_l_fp8 carry a `parity` object: one 640x640 image pair through the record's final weights, train-mode heads and six
                parameter gradients against fp32 CPU autograd through the oracle -- relative L2, cosine)
  train_l_fp8   BASELINE configs[4]: YOLOPoint-l optimizer step, 16 samples per GPU (bs 128 over 8 GPUs), data parallel over all N ranks, fp8 Conv
                operands (e4m3 x e4m3 forward, e5m2 x e4m3 dgrad, bf16 storage / BatchNorm / weight gradients); at N = 1 `bf16_ms_per_step` = the
                same step in bf16
  frame         (N = 1) BASELINE configs[3]: YOLOPoint-l, one 1280x1280 frame end to end (forward + keypoint decode / NMS + box NMS on 100 800
                rows + box-mask filter + descriptor sampling + MNN matching against the previous frame)
  v52           (N = 1) the model the reference's shipped inference config selects (configs/kitti_inference.yaml:2): YOLOPointv52-s, batch 8,
                640x640, f16, hipGraph replay, with its own `parity` against the oracle at that shape
`cpu_baseline` objects: top level (oracle forward), `train.cpu_baseline` (oracle forward + autograd backward + torch.optim.Adam of one image
pair, fp32), `frame.cpu_baseline` (oracle forward of one 1280x1280 frame + the oracle's sequential post-processing on the same planted heads).
`--only infer|train|frame` restricts the run; `--mode train|frame|export` prints that workload as the top-level record (round-1 CLI).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}     # MI355X_MICROARCH.md dense MFMA peaks (fp8: the block-scaled K = 64 / 128 forms)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--version", default="s")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--no-graph", action="store_true", help="replay the plan eagerly instead of through a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", default="", help="write a per-launch table (us, TFLOP/s, GB/s) to this path")
    ap.add_argument("--postproc", action="store_true", help="also time the post-processing kernels on planted head outputs")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "frame", "export"],
                    help="infer (default, BASELINE.json configs[1] + sub-records) or one workload as the top-level record")
    ap.add_argument("--only", default="", help="comma list of sub-records to run beside the top level: train,train64,frame,fp8,v52,bs1 (default: all)")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--train-warmup", type=int, default=3)
    ap.add_argument("--frame-steps", type=int, default=60)
    ap.add_argument("--gas", type=int, default=1, help="--mode train: micro-batches per optimizer step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL over xGMI; gloo only with --dry)")
    ap.add_argument("--dry", action="store_true", help="launcher check: every rank joins the group, runs a stub step (no GPU, no model), rank 0 prints the line")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 on a box with fewer than N GPUs: all ranks share the visible GPU(s) and exchange through gloo (RCCL refuses two ranks on one "
                         "device).  Exercises the whole N > 1 control flow of this file -- replica timing, shared autotuning, data-parallel training with "
                         "the real bucketed gradient all-reduce -- on real kernels; the record is marked `rehearsal` and its throughput means nothing")
    a = ap.parse_args()
    if a.backend == "gloo" and not (a.dry or a.rehearsal):
        ap.error("--backend gloo is for --dry (launcher check) and --rehearsal (ranks sharing a GPU) only: the product has no CPU path")
    if a.rehearsal:
        a.backend = "gloo"
    return a


def self_launch(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): become `torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` -- one rank per GPU, the form the reference is
    started in (`accelerate launch src/train.py`, README.md:76; train.py:38-46,174).  The process image is replaced (exec), so rank 0's ONE
    JSON line reaches this process's stdout unchanged.  Fails loudly when the box has fewer than N GPUs (never a silent N = 1 run)."""
    if not a.dry:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if a.rehearsal and have < 1:
            sys.stderr.write("bench.py: --rehearsal still needs one visible GPU\n")
            sys.exit(2)
        if have < a.gpus and not a.rehearsal:
            sys.stderr.write(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs, this box has {have} "
                             f"(one rank per GPU over RCCL; there is no CPU path and no oversubscription of a GPU)\n")
            sys.exit(2)
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = str(s_.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: launching " + " ".join(cmd) + "\n")
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def dry_run(a, rank, world):
    """--dry: what the launcher test checks without a GPU -- N ranks came up, joined ONE process group, ran the timed-region protocol of the
    real records (dp.timed_region: warm-up, barrier, K stub steps with an all-reduce each, barrier, MAX over ranks) and rank 0 alone printed
    the line with n_gpus = N and the list of ranks that reported."""
    import torch.distributed as dist
    from yolopoint_amd.dp import timed_region
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend)
    buf = torch.zeros(1024)

    def step():
        buf.fill_(float(rank + 1))
        if world > 1:
            dist.all_reduce(buf)

    def reduce_max(t):
        if world == 1:
            return t
        x = torch.tensor([t], dtype=torch.float64)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        return float(x.item())
    wall = timed_region(step, a.steps, a.warmup, lambda: None, (dist.barrier if world > 1 else (lambda: None)), reduce_max)
    ranks = [None] * world
    if world > 1:
        dist.all_gather_object(ranks, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(), "sum": float(buf[0])})
    else:
        ranks = [{"rank": 0, "local_rank": 0, "pid": os.getpid(), "sum": float(buf[0])}]
    if rank == 0:
        emit(json.dumps({"metric": "launcher dry run (stub step, no GPU work)", "value": round(a.steps * world / wall, 1), "unit": "stub steps/s", "n_gpus": world,
                         "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                         "vs_baseline": None, "dtype": "none", "data": "none", "dry": True, "backend": a.backend,
                         "config": {"workload": "launcher check", "ranks_reported": [r["rank"] for r in ranks], "gpus_arg": a.gpus,
                                    "allreduce_sum": ranks[0]["sum"], "expected_sum": world * (world + 1) / 2},
                         "ranks": ranks}))
    if world > 1:
        dist.destroy_process_group()


def build_model(version, dtype, dev):
    from yolopoint_amd.utils.synthetic import make_model
    m, sd = make_model(version, 1234, dtype=dtype)
    m = m.to(dev)
    m.fuse()                          # inference path: BN folded (reference demo.py:48-49)
    m.model.static_outputs = True     # outputs live in the plan's static buffers (graph replay semantics)
    return m, sd


def cpu_baseline(version, B, S, budget_s=14.0, gpu_outs=None):
    """Oracle forward on the host cores (PyTorch-CPU fp32, eval).  Bounded sample: the thread count is picked by a short probe
    (oversubscribing a 256-thread host makes ATen's small convolutions crawl), then 3 warm-ups and up to 10 timed iterations or
    `budget_s` seconds of the same workload.  With `gpu_outs` (the timed plan's head outputs for the same weights and input) the
    oracle's outputs are also the reference of the `parity` record."""
    from oracle import net_oracle                     # the CPU baseline IS the oracle's forward; nothing else in this file touches oracle/
    from yolopoint_amd.utils.synthetic import NAMES80, layout_of, GAIN
    from yolopoint_amd import models
    cores = os.cpu_count() or 1
    layout = layout_of(models.Model(names=NAMES80, version=version))
    sd = net_oracle.synth_state_dict(layout, 1234, gain=GAIN.get(version, 1.6))
    Bc = min(B, 8)
    x = net_oracle.synth_image(B, 3, S, S, 1234)[:Bc]
    best_t, best_n = None, None
    with torch.no_grad():
        for n in sorted({min(cores, c) for c in (16, 32, 64)}):
            torch.set_num_threads(n)
            net_oracle.yolopoint_forward(sd, x[:1], version)                     # warm-up
            t0 = time.perf_counter()
            net_oracle.yolopoint_forward(sd, x[:2] if Bc >= 2 else x, version)
            dt = time.perf_counter() - t0
            if best_t is None or dt < best_t:
                best_t, best_n = dt, n
        torch.set_num_threads(best_n)
        for _ in range(3):                                                        # warm-ups at full batch (>= 3)
            ref = net_oracle.yolopoint_forward(sd, x, version)
        times = []
        t_start = time.perf_counter()
        while len(times) < 10 and (len(times) < 3 or (time.perf_counter() - t_start) < budget_s):
            t0 = time.perf_counter()
            net_oracle.yolopoint_forward(sd, x, version)
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    base = {"value": round(Bc / med, 2), "unit": "images/s", "cores": best_n, "kind": "port",
            "threads_note": f"{best_n} threads = the fastest of 16 / 32 / 64 on this {cores}-thread host (SURVEY 8d says os.cpu_count(): oversubscribing ATen's small "
                            f"convolutions is slower, so the probe favours the CPU)",
            "sample": f"oracle (PyTorch-CPU fp32, eval) forward of YOLOPoint-{version} batch {Bc} {S}x{S}: {best_n} threads "
                      f"(best of 16/32/64 on a {cores}-thread host), 3 warm-ups + {len(times)} timed iterations, median"}
    parity = None
    if gpu_outs is not None and Bc == B:
        def rel(a, b):
            a, b = a.detach().double().cpu(), b.detach().double()
            return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30))
        parity = {"against": "oracle fp32 CPU forward, same weights and input as the timed plan", "dtype": gpu_outs["dtype"]}
        for k in ("semi", "desc"):
            parity[k + "_max_rel"], parity[k + "_rel_l2"] = (round(v, 6) for v in rel(gpu_outs[k], ref[k]))
        parity["pred_rel_l2"] = round(rel(gpu_outs["pred"], ref["objects"][0])[1], 6)
        parity["raw_levels_rel_l2"] = [round(rel(a, b)[1], 6) for a, b in zip(gpu_outs["xs"], ref["objects"][1])]
        parity["keypoint_cell_argmax_agreement"] = round(float((gpu_outs["semi"].argmax(1).cpu() == ref["semi"].argmax(1)).float().mean()), 6)
        # the floor of 16-bit inference itself: the oracle with fp16 storage of activations / BN-folded filters and fp32 accumulation
        # (PyTorch's half-precision arithmetic on the CPU) against the fp32 oracle -- the HIP path is held to 1.15x of it in relative L2
        with torch.no_grad(), net_oracle.half_storage(torch.float16 if gpu_outs["dtype"] == "f16" else torch.bfloat16):
            flo = net_oracle.yolopoint_forward(net_oracle.fused_state_dict(sd), x, version)
        fl = {"semi": rel(flo["semi"], ref["semi"])[1], "desc": rel(flo["desc"], ref["desc"])[1], "pred": rel(flo["objects"][0], ref["objects"][0])[1],
              "raw_levels": [rel(a, b)[1] for a, b in zip(flo["objects"][1], ref["objects"][1])]}
        parity["floor_rel_l2"] = {k: ([round(v, 6) for v in fl[k]] if isinstance(fl[k], list) else round(fl[k], 6)) for k in fl}
        parity["ratio_to_floor"] = {"semi": round(parity["semi_rel_l2"] / fl["semi"], 3), "desc": round(parity["desc_rel_l2"] / fl["desc"], 3),
                                    "pred": round(parity["pred_rel_l2"] / fl["pred"], 3),
                                    "raw_levels": [round(a / b, 3) for a, b in zip(parity["raw_levels_rel_l2"], fl["raw_levels"])]}
        parity["floor_argmax_agreement"] = round(float((flo["semi"].argmax(1) == ref["semi"].argmax(1)).float().mean()), 6)
        parity["bars"] = ("tests/test_gpu_bench_shapes.py: every head tensor's relative L2 <= 1.15 x floor_rel_l2 (max-abs <= 3.0 x), argmax mismatches only on "
                          "reference near-ties within one 16-bit step; the f32 path meets the north-star 1e-3 of max|ref| with bit-exact argmax")
    return base, parity


def main(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # a launcher's WORLD_SIZE and --gpus must say the same thing: a mismatch means N ranks were asked for and another number started
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)
    if a.dry:
        return dry_run(a, rank, world)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend)
    if a.rehearsal:
        local = local % max(torch.cuda.device_count(), 1)          # ranks share the visible GPU(s)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from yolopoint_amd import _hip
    _hip.require_gpu()
    if a.mode == "train":
        return bench_train(a, rank, world, dev)
    if a.mode == "frame":
        return bench_frame(a, dev)
    if a.mode == "export":
        return bench_export(a, dev)
    m, _ = build_model(a.version, a.dtype, dev)
    net = m.model
    B, S = a.batch, a.size
    from yolopoint_amd.utils.synthetic import synth_image
    x = synth_image(B, 3, S, S, 1234 + rank).to(dev)
    from yolopoint_amd.plan import pack_input

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan, img, outs = net.build_plan(B, S, S, dev, graph=not a.no_graph)

        def step():
            net.run_plan(plan, img, x)

        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        gpu_ms = e0.elapsed_time(e1)

        # ---- per-launch durations (HIP events between launches on this stream), median of 5 passes
        passes = [plan.profile() for _ in range(5)]
        nops = len(passes[0])
        per_op = [sorted(p[i] for p in passes)[2] for i in range(nops)]
        recs_extra, per_op_extra = [], []
        if plan.stem_launch is not None:          # the fused stem runs outside the plan: time it the same way
            ts = []
            for _ in range(5):
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record(); plan.stem_launch(x); s1.record(); s1.synchronize()
                ts.append(s0.elapsed_time(s1))
            recs_extra, per_op_extra = [plan.stem_record], [sorted(ts)[2]]

    if world > 1:
        t = torch.tensor([wall], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    recs = recs_extra + list(plan.records)
    per_op = per_op_extra + per_op
    conv_ms = sum(ms for ms, r in zip(per_op, recs) if r.kind == "conv")
    conv_flops = sum(r.flops for r in recs if r.kind == "conv")
    conv_bytes = sum(r.bytes for r in recs if r.kind == "conv")
    other_ms = sum(ms for ms, r in zip(per_op, recs) if r.kind != "conv")
    n_conv = sum(1 for r in recs if r.kind == "conv")
    serial_achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # Two-lane plans overlap their launches (the heads run beside the PAN chain), so the SUM of the isolated launch durations no longer
    # is the time the convolution kernels occupy the chip; the timed step is (it also contains the two non-convolution launches, so the
    # rate derived from it is a lower bound of the convolution kernels' own).  One-lane plans keep the per-launch sum.
    lanes = bool(getattr(plan, "has_lanes", False))
    # (the contract's ms_per_step -- wall clock, max over ranks -- not the HIP-event time, so that frac x peak x ms_per_step reproduces the FLOP count)
    achieved = conv_flops / (wall / a.steps) / 1e12 if lanes else serial_achieved
    peak = PEAK_TFLOPS[a.dtype]
    # HBM traffic of the conv kernel comes from separate rocprofv3 --pmc passes of this same command (PMC counters
    # cannot be read in-process); profiles/conv_traffic.json holds the latest committed measurement.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")
    if os.path.exists(tpath) and (a.version, B, S, a.dtype) == ("s", 8, 640, "f16"):
        try:
            traffic = round(json.load(open(tpath))["hbm_bytes_per_launch"])
        except Exception:
            traffic = None

    if a.layers and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.layers)) or ".", exist_ok=True)
        with open(a.layers, "w") as f:
            f.write(f"# per-launch profile, YOLOPoint-{a.version} B={B} {S}x{S} {a.dtype}; eager launches, HIP events, median of 5\n")
            f.write(f"{'op':52s} {'kind':7s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>9s} {'TFLOP/s':>9s} {'GB/s(alg)':>10s}\n")
            for ms, r in zip(per_op, recs):
                tf = r.flops / (ms * 1e-3) / 1e12 if ms > 0 else 0
                gb = r.bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0
                f.write(f"{r.name:52s} {r.kind:7s} {r.M:8d} {r.N:5d} {r.K:5d} {ms * 1e3:9.1f} {tf:9.1f} {gb:10.0f}\n")
            f.write(f"# conv: {n_conv} launches {conv_ms * 1e3:.1f} us, {conv_flops / 1e9:.2f} GFLOP -> {achieved:.1f} TFLOP/s; "
                    f"other ops {other_ms * 1e3:.1f} us; graph step {gpu_ms / a.steps * 1e3:.1f} us\n")

    # backbone conv stack (Conv1 .. SPPooling, SURVEY 8d F_bb) separately
    bb_names = ("Conv1", "Conv2", "Bottleneck1", "Conv3", "Bottleneck2", "Conv4", "Bottleneck3", "Conv5", "Bottleneck4", "SPPooling")
    # (a fused launch is named after its parts -- "Conv1+Conv2+Bottleneck1.cv1+cv2...", "Bottleneck2.cv1+cv2", "...m.0.cv1>cv2>cv3": the module of
    # its FIRST part decides; every fused launch lies inside one stage of the network)
    import re
    bb = [(ms, r) for ms, r in zip(per_op, recs) if r.kind == "conv" and re.split(r"[.+>]", r.name)[0] in bb_names]
    bb_ms, bb_flops = sum(ms for ms, _ in bb), sum(r.flops for _, r in bb)
    gpu_outs = None
    if world == 1 and not a.no_cpu_baseline:
        net.run_plan(plan, img, x)
        torch.cuda.synchronize()
        dch = net.ConvDesc.out_channels if hasattr(net, "ConvDesc") else net._desc_channels
        gpu_outs = {"dtype": a.dtype, "semi": outs["semi"].buf.t[..., :65].permute(0, 3, 1, 2).float().cpu(),
                    "desc": outs["desc"].buf.t[..., :dch].permute(0, 3, 1, 2).float().cpu(), "pred": outs["z"].float().cpu(),
                    "xs": [t.float().cpu() for t in outs["xs"]]}
    imgs = B * a.steps * world
    out = None
    if rank == 0:
        out = {
            "metric": "images/sec at 640x640 (YOLOPoint-s inference, bs=8, fp16)" if (a.version, B, S, a.dtype) == ("s", 8, 640, "f16")
            else f"images/sec at {S}x{S} (YOLOPoint-{a.version} inference, bs={B}, {a.dtype})",
            "value": round(imgs / wall, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(wall / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            **({"rehearsal": "ranks share one GPU and exchange through gloo: control-flow check only, the numbers are not throughput"} if a.rehearsal else {}),
            "config": {"workload": f"BASELINE.json configs[1]: YOLOPoint-{a.version} inference forward (backbone + heads + Detect decode), batch {B}/GPU, "
                                   f"{S}x{S}, {a.dtype} / fp32 accumulate, BN folded, inputs resident in HBM",
                       "per_gpu_batch": B, "global_batch": B * world, "image": [S, S],
                       "parallelism": "replicas" if world > 1 else "single",
                       "launch": ("two streams (main lane + side lane), eager launches" if lanes and not plan.graph else
                                  ("eager" if a.no_graph else "hipGraph")), "ops_per_step": len(per_op) + (0 if plan.stem_launch else 1),
                       "scaling_records": "value = inference replicas (weak); train.value = data-parallel training over the same N ranks (weak, 8 samples/GPU)"},
            "gpu_ms_per_step_events": round(gpu_ms / a.steps, 4),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": "profiles/conv_traffic.json: the committed rocprofv3 --pmc measurement of this command on the profile box (PMC counters cannot be read in-process)",
                         "algorithmic_bytes_per_launch": round(conv_bytes / max(n_conv, 1)),
                         "kernel": "the convolution kernels: conv_igemm / conv_mma8 / conv3x3_halo / bottleneck_halo / stem_conv (all instantiations)",
                         "achieved_from": ("conv FLOP per step / ms_per_step (wall clock of the timed region; the launches of the two lanes overlap)" if lanes else
                                           "conv FLOP per step / sum of the conv launch durations (HIP events, one lane)"),
                         "serial_launch_sum": {"conv_us_per_step": round(conv_ms * 1e3, 1), "achieved": round(serial_achieved, 2),
                                               "frac": round(serial_achieved / peak, 4),
                                               "note": "every launch alone on the chip, HIP events between eager launches on one stream"},
                         "launches_per_step": n_conv, "conv_us_per_step": round(conv_ms * 1e3, 1),
                         "algorithmic_gflop_per_step": round(conv_flops / 1e9, 3),
                         "algorithmic_hbm_frac": round(conv_bytes / (conv_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if conv_ms > 0 else None,
                         "whole_step_tflops": round(conv_flops / (gpu_ms / a.steps * 1e-3) / 1e12, 2),
                         "whole_step_frac": round(conv_flops / (gpu_ms / a.steps * 1e-3) / 1e12 / peak, 4),
                         "backbone": {"layers": "Conv1..SPPooling", "gflop_per_step": round(bb_flops / 1e9, 3), "us_per_step": round(bb_ms * 1e3, 1),
                                      "achieved": round(bb_flops / (bb_ms * 1e-3) / 1e12, 2) if bb_ms > 0 else None,
                                      "frac": round(bb_flops / (bb_ms * 1e-3) / 1e12 / peak, 4) if bb_ms > 0 else None},
                         "mfma_busy": mfma_busy_record()},
        }
        if a.postproc:
            out["postproc"] = bench_postproc(dev)
    # ---- sub-records (every rank takes part in the data-parallel training; the rest is rank 0, N = 1)
    only = set(k for k in a.only.split(",") if k) or {"train", "train64", "frame", "fp8", "v52", "bs1"}
    a.cpu_threads = None
    parity = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"], parity = cpu_baseline(a.version, B, S, gpu_outs=gpu_outs)
        a.cpu_threads = out["cpu_baseline"]["cores"]
        if parity is not None:
            out["parity"] = parity
    del plan, img, outs
    net.__dict__.pop("_plans", None)
    torch.cuda.empty_cache()
    if "fp8" in only:
        # (first among the sub-records: measured after the -s training records, one of the two -l steps came out ~2 ms slower -- which
        # one depended on what had run before; on a fresh process both agree with their stand-alone runs)
        # BASELINE configs[4]: YOLOPoint-l, 16 samples per GPU (bs 128 over 8 GPUs), fp8 Conv operands, data parallel over all N ranks;
        # at N = 1 the bf16 step of the same model is timed beside it
        torch.cuda.empty_cache()
        rec = run_train(a, rank, world, dev, "l", 16, max(8, a.train_steps // 2), 3, gas=1, dtype="fp8")
        if world == 1:
            ref = run_train(a, rank, world, dev, "l", 16, max(8, a.train_steps // 2), 3, gas=1, dtype="bf16", parity=False)
            rec["bf16_ms_per_step"] = ref["ms_per_step"]
        if rank == 0:
            out["train_l_fp8"] = rec
    if "train" in only:
        rec = run_train(a, rank, world, dev, a.version, 8, a.train_steps, a.train_warmup, gas=1)
        if rank == 0:
            out["train"] = rec
    if world == 1 and "train64" in only:
        # train.py:38-43: gas = max(round(64 / (train_batch_size * devices)), 1) -- with train_batch_size 64 on one device that is ONE batch
        # of 64 samples per optimizer step (BatchNorm statistics over the 128 images of the batch); train_batch_size 8 reaches the nominal
        # batch with 8 micro-batches (`gas8_ms_per_step`).  One pass over 64 samples: 46.1 ms, eight passes over 8: 59.0 ms (same box).
        torch.cuda.empty_cache()
        cpu_threads, a.cpu_threads = a.cpu_threads, None          # (the CPU baseline of the training step rides with the `train` record)
        rec64 = run_train(a, rank, world, dev, a.version, 64, max(3, a.train_steps // 4), 1, gas=1)
        a.cpu_threads = cpu_threads
        torch.cuda.empty_cache()
        rec64["gas8_ms_per_step"] = run_train(a, rank, world, dev, a.version, 8, max(3, a.train_steps // 4), 1, gas=8, parity=False)["ms_per_step"]
        out["train_bs64"] = rec64
    if world == 1 and "frame" in only:
        torch.cuda.empty_cache()
        out["frame"] = run_frame(dev, "l", 1280, a.dtype, a.frame_steps, max(3, a.frame_steps // 6), cpu_threads=a.cpu_threads)
    if rank != 0:
        import torch.distributed as dist          # (every rank leaves the group itself; no collective behind the last sub-record, so no barrier is needed)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if world == 1 and "v52" in only:
        torch.cuda.empty_cache()
        out["v52"] = run_v52(dev, 100, 10, a.cpu_threads or 16, a.cpu_threads is not None)
    if world == 1 and "bs1" in only:
        torch.cuda.empty_cache()
        out["infer_bs1"] = run_infer_bs1(dev, a.dtype, 300, 30)
    # the numbers the review compares, once more at the FRONT of the line (a consumer that keeps only the head or only the tail of a long line
    # still sees them): ms per step of every sub-record
    summ = {"infer_bs8_ms": out["ms_per_step"], "roofline_frac": out["roofline"]["frac"]}
    for k_ in ("train", "train_bs64", "train_l_fp8", "frame", "v52", "infer_bs1"):
        if k_ in out and isinstance(out[k_], dict) and "ms_per_step" in out[k_]:
            summ[k_ + "_ms"] = out[k_]["ms_per_step"]
    if "train_l_fp8" in out and "bf16_ms_per_step" in out["train_l_fp8"]:
        summ["train_l_bf16_ms"] = out["train_l_fp8"]["bf16_ms_per_step"]
        summ["fp8_over_bf16"] = round(out["train_l_fp8"]["ms_per_step"] / out["train_l_fp8"]["bf16_ms_per_step"], 4)
    # ... and as short numeric keys inside `config` / `roofline`, the two objects a record store keeps verbatim
    short = {}
    for key, name in (("infer_bs1", "bs1"), ("train", "train"), ("train_bs64", "train_bs64"), ("train_l_fp8", "l_fp8"), ("frame", "frame"), ("v52", "v52")):
        r_ = out.get(key)
        if isinstance(r_, dict) and "ms_per_step" in r_:
            short[name + "_ms"] = r_["ms_per_step"]
            if isinstance(r_.get("roofline"), dict):
                short[name + "_frac"] = r_["roofline"].get("frac")
            if isinstance(r_.get("parity"), dict) and "grad_rel_l2_median" in r_["parity"]:
                short[name + "_grad_l2"] = r_["parity"]["grad_rel_l2_median"]
    if "train_l_bf16_ms" in summ:
        short["l_bf16_ms"], short["fp8_over_bf16"] = summ["train_l_bf16_ms"], summ["fp8_over_bf16"]
    if isinstance(out["roofline"].get("backbone"), dict):
        short["backbone_frac"], short["backbone_gflop"] = out["roofline"]["backbone"]["frac"], out["roofline"]["backbone"]["gflop_per_step"]
    out["config"].update(short)
    out["roofline"].update(short)
    head = {k_: out[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}
    head["summary"] = summ
    head.update({k_: v_ for k_, v_ in out.items() if k_ not in head})
    emit(json.dumps(head))
    if __import__("torch").distributed.is_initialized():
        try:
            __import__("torch").distributed.destroy_process_group()
        except Exception:
            pass


def mfma_busy_record():
    """MFMA utilisation of the timed plan from the PMC counters (cannot be read in-process): the latest committed measurement,
    profiles/mfma_busy.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of this command, tools/profile_collect.py)."""
    path = os.path.join(ROOT, "profiles", "mfma_busy.json")
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None



def train_traffic_record(version, batch, dtype):
    """HBM bytes of one optimizer step from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    `bench.py --mode train`, FETCH_SIZE x2 per MI355X_MICROARCH.md): the latest committed measurement, profiles/train_traffic.json."""
    path = os.path.join(ROOT, "profiles", "train_traffic.json")
    try:
        d = json.load(open(path))
        return d.get(f"{version}_{batch}_{dtype}")
    except Exception:
        return None


def cpu_train_baseline(version, S, threads, budget_s=18.0):
    """The reference optimizer step on the host cores, bounded: ONE image pair (B = 1 sample: two train-mode forwards through the oracle,
    fp32), autograd backward, torch.optim.Adam over all parameters.  The three loss heads are replaced by seeded projections of the
    outputs (oracle.net_oracle.projected_loss: the reference's losses are < 1 % of the step's FLOP); the warped pass back-propagates its
    keypoint / descriptor outputs only, as in src/train.py:208-245."""
    from oracle import net_oracle
    from yolopoint_amd.utils.synthetic import NAMES80, layout_of, GAIN
    from yolopoint_amd import models
    torch.set_num_threads(threads)
    layout = layout_of(models.Model(names=NAMES80, version=version))
    sd = net_oracle.synth_state_dict(layout, 1234, gain=GAIN.get(version, 1.6))
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    params = [v for v in leaf.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    x, xw = net_oracle.synth_image(1, 3, S, S, 1), net_oracle.synth_image(1, 3, S, S, 2)
    proj = None

    def step():
        nonlocal proj
        opt.zero_grad(set_to_none=True)
        o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
        ow = net_oracle.yolopoint_forward(leaf, xw, version, training=True, stats={})
        if proj is None:
            proj = net_oracle.output_projections(o, 3)
        loss = net_oracle.projected_loss(o, proj) + (ow["semi"] * proj["semi"]).sum() * 0.01 + (ow["desc"] * proj["desc"]).sum()
        loss.backward()
        opt.step()
    step()
    times, t_start = [], time.perf_counter()
    while len(times) < 5 and (len(times) < 2 or (time.perf_counter() - t_start) < budget_s):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(2.0 / med, 3), "unit": "images/s (an image pair counts as 2 images)", "cores": threads, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32) optimizer step of YOLOPoint-{version} on ONE {S}x{S} image pair (the GPU record runs 8-16 pairs per step): two "
                      f"train-mode forwards, autograd backward (warped pass: keypoint / descriptor heads only), torch.optim.Adam; losses = seeded output "
                      f"projections; {threads} threads (the count the forward probe picked), 1 warm-up + {len(times)} timed steps, median"}


def cpu_frame_baseline(version, S, threads, semis, preds, budget_s=25.0):
    """One frame on the host cores: the oracle's fp32 forward of one SxS image + the oracle's sequential post-processing (keypoint decode,
    greedy grid NMS, box NMS, box-mask filter, descriptor sampling) on the SAME planted heads the GPU record uses + mutual-NN matching
    against the previous frame."""
    from oracle import net_oracle, postproc_oracle as po
    from yolopoint_amd.utils.synthetic import NAMES80, layout_of, GAIN
    from yolopoint_amd import models
    torch.set_num_threads(threads)
    layout = layout_of(models.Model(names=NAMES80, version=version))
    sd = net_oracle.synth_state_dict(layout, 1234, gain=GAIN.get(version, 1.6))
    fused = sd
    prev = [None]
    times, t_start = [], time.perf_counter()
    i = 0
    with torch.no_grad():
        while len(times) < 4 and (len(times) < 2 or (time.perf_counter() - t_start) < budget_s):
            x = net_oracle.synth_image(1, 3, S, S, 100 + i)
            t0 = time.perf_counter()
            o = net_oracle.yolopoint_forward(fused, x, version)
            k = i % len(semis)
            r = po.frontend_postprocess(semis[k][0].numpy(), o["desc"][0].numpy(), preds[k][0].numpy())
            desc = r[1] if isinstance(r, tuple) else r["desc"]
            if prev[0] is not None and desc is not None and desc.shape[1] and prev[0].shape[1]:
                po.nn_match_two_way(desc, prev[0], 0.7)
            prev[0] = desc
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)          # (the first frame is the warm-up and has no previous frame to match)
            i += 1
    med = sorted(times)[len(times) // 2]
    return {"value": round(1.0 / med, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle forward (PyTorch-CPU fp32) of one {S}x{S} frame through YOLOPoint-{version} + oracle/postproc_oracle.py on the planted heads of the "
                      f"GPU record (sequential greedy NMS loops as the reference runs them) + matching; {threads} threads, 1 warm-up + {len(times)} timed frames, median"}


def run_infer_bs1(dev, dtype, steps, warmup):
    """BASELINE.json `metric` quotes "infer bs=1" at 640x640: YOLOPoint-s, ONE image per forward (latency and images/s), same plan machinery
    as the top-level record (two lanes, eager two-stream replay)."""
    m, _ = build_model("s", dtype, dev)
    net = m.model
    from yolopoint_amd.utils.synthetic import synth_image
    S = 640
    x = synth_image(1, 3, S, S, 4321).to(dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan, img, outs = net.build_plan(1, S, S, dev, graph=True)
        for _ in range(warmup):
            net.run_plan(plan, img, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.run_plan(plan, img, x)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    conv_flops = sum(r.flops for r in plan.records if r.kind == "conv") + (plan.stem_record.flops if plan.stem_launch else 0)
    tf = conv_flops / (wall / steps) / 1e12
    rec = {"metric": f"images/sec at 640x640 (YOLOPoint-s inference, bs=1, {dtype})", "value": round(steps / wall, 1), "unit": "images/s",
           "ms_per_step": round(wall / steps * 1e3, 4), "latency_ms": round(wall / steps * 1e3, 4), "steps": steps, "warmup": warmup, "dtype": dtype,
           "config": {"workload": "BASELINE.json metric, 'infer bs=1': YOLOPoint-s forward of ONE 640x640 image (backbone + heads + Detect decode), BN folded, "
                                  "input resident in HBM", "launch": "two streams (main lane + side lane), eager launches" if plan.has_lanes and not plan.graph else "hipGraph",
                      "ops_per_step": plan.num_ops() + (1 if plan.stem_launch else 0)},
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS[dtype], 4), "traffic": None,
                        "algorithmic_gflop_per_step": round(conv_flops / 1e9, 3),
                        "note": "conv FLOP of one image / the timed step: ~50 dependent launches of a few microseconds each -- launch latency, not the MFMA rate, bounds one image"}}
    del plan, img, outs
    net.__dict__.pop("_plans", None)
    return rec


def run_v52(dev, steps, warmup, threads, with_cpu):
    """YOLOPointv52-s (the model reference configs/kitti_inference.yaml:2 selects), batch 8, 640x640, f16, hipGraph: images/s + parity at
    that shape against the oracle's fp32 forward."""
    from yolopoint_amd.utils.synthetic import make_model, synth_image
    B, S = 8, 640
    m, sd = make_model("s", 1234, dtype="f16", model_name="YOLOPointv52")
    m = m.to(dev)
    m.fuse()
    m.model.static_outputs = True
    net = m.model
    x = synth_image(B, 3, S, S, 1234).to(dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan, img, outs = net.build_plan(B, S, S, dev, graph=True)
        for _ in range(warmup):
            net.run_plan(plan, img, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.run_plan(plan, img, x)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        per_op = plan.profile()
    recs = list(plan.records)
    conv_ms = sum(ms for ms, r in zip(per_op, recs) if r.kind == "conv")
    conv_flops = sum(r.flops for r in recs if r.kind == "conv")
    if getattr(plan, "has_lanes", False):          # two lanes: the launches overlap, the timed step is the time the kernels occupy the chip
        conv_ms = wall / steps * 1e3
        if plan.stem_launch:
            conv_flops += plan.stem_record.flops
    rec = {"metric": "images/sec at 640x640 (YOLOPointv52-s inference, bs=8, fp16)", "value": round(B * steps / wall, 1), "unit": "images/s",
           "ms_per_step": round(wall / steps * 1e3, 4), "steps": steps, "warmup": warmup, "dtype": "f16",
           "config": {"workload": "reference configs/kitti_inference.yaml:2 model (YOLOPointv52-s), batch 8, 640x640, BN folded, "
                                  + ("two-lane eager replay" if getattr(plan, "has_lanes", False) and not plan.graph else "hipGraph replay"),
                      "ops_per_step": len(per_op)},
           "roofline": {"bound": "mfma", "achieved": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else None, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(conv_flops / (conv_ms * 1e-3) / 1e12 / 2500.0, 4) if conv_ms > 0 else None,
                        "traffic": train_traffic_record("v52_s", 8, "f16"),
                        "algorithmic_gflop_per_step": round(conv_flops / 1e9, 3),
                        "note": "conv FLOP per step / the timed step (two lanes overlap)" if getattr(plan, "has_lanes", False) else "conv launches of the plan (eager, HIP events) against their algorithmic FLOP"}}
    if with_cpu:
        from oracle import net_oracle
        torch.set_num_threads(threads)
        net.run_plan(plan, img, x)
        torch.cuda.synchronize()
        dch = net._desc_channels if hasattr(net, "_desc_channels") else outs["desc"].buf.t.shape[-1]
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = net_oracle.yolopointv52_forward(sd, x.cpu(), "s")
            dt = time.perf_counter() - t0

        def rel(a_, b_):
            a_, b_ = a_.detach().double().cpu(), b_.detach().double()
            return round(float((a_ - b_).norm() / b_.norm().clamp_min(1e-30)), 6)
        semi = outs["semi"].buf.t[..., :65].permute(0, 3, 1, 2).float()
        desc = outs["desc"].buf.t[..., :ref["desc"].shape[1]].permute(0, 3, 1, 2).float()
        rec["parity"] = {"against": "oracle.net_oracle.yolopointv52_forward (fp32 CPU), same weights and input", "semi_rel_l2": rel(semi, ref["semi"]),
                         "desc_rel_l2": rel(desc, ref["desc"]), "pred_rel_l2": rel(outs["z"].float(), ref["objects"][0]),
                         "keypoint_cell_argmax_agreement": round(float((semi.argmax(1).cpu() == ref["semi"].argmax(1)).float().mean()), 6)}
        rec["cpu_baseline"] = {"value": round(B / dt, 2), "unit": "images/s", "cores": threads, "kind": "port",
                               "sample": f"ONE oracle forward (PyTorch-CPU fp32) of YOLOPointv52-s batch 8 640x640, {threads} threads (also the parity reference)"}
    del plan, img, outs
    net.__dict__.pop("_plans", None)
    torch.cuda.empty_cache()
    return rec

PARITY_TENSORS = ["model.Conv2.conv.weight", "model.Bottleneck2.cv3.conv.weight", "model.SPPooling.cv2.conv.weight", "model.Bottleneck6.cv1.conv.weight",
                  "model.ConvDesc.weight", "model.Detect.m.1.weight"]


def train_parity(m, version, dev, fp8, threads, S=640, seed=77):
    """`parity` of a training record, measured in this process on the weights the timed steps ended with: ONE image pair (a batch of two
    640x640 images) through the model's train-mode forward and native backward against PyTorch-CPU fp32 autograd through the oracle on the same
    weights and input (the loss is the seeded projection of the three heads the golden backward fixture uses, oracle.net_oracle.projected_loss).
    Head outputs: relative L2; gradients of six parameter tensors spread over the network (backbone, SPPF, PAN, descriptor head, Detect):
    relative L2 and cosine.  fp8 records carry the numbers of the fp8 step AND of the same weights stepped in bf16 (8-bit rounding is
    discontinuous: the fp32 oracle is the common yardstick, the rule-for-rule comparison is tests/test_gpu_fp8.py's fake-quantised oracle)."""
    from oracle import net_oracle                     # checker only (behind the timed region)
    from yolopoint_amd.models.common import invalidate_packed_weights
    torch.set_num_threads(threads)
    sd = {k: (v.detach().float().cpu().clone() if v.dtype.is_floating_point else v.detach().cpu().clone()) for k, v in m.state_dict().items()}
    x = net_oracle.synth_image(2, 3, S, S, seed)
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    o = net_oracle.yolopoint_forward(leaf, x, version, training=True, stats={})
    proj = net_oracle.output_projections(o, seed)
    net_oracle.projected_loss(o, proj).backward()

    def rel(a_, b_):
        a_, b_ = a_.detach().double().cpu(), b_.detach().double()
        return float((a_ - b_).norm() / b_.norm().clamp_min(1e-30))

    def hip_pass(use_fp8):
        net = m.model
        was = bool(getattr(net, "fp8_train", False))
        net.fp8_train = bool(use_fp8)
        bn0 = [b.detach().clone() for b in m.buffers()]
        try:
            for _ in range(3 if use_fp8 else 1):          # fp8: the scales lag one pass behind -- two calibration passes on the same input
                m.zero_grad(set_to_none=True)
                with torch.no_grad():
                    for b, s0 in zip(m.buffers(), bn0):
                        b.copy_(s0)
                out = m(x.to(dev))
                net_oracle.projected_loss(out, proj, dev).backward()
                if use_fp8:
                    invalidate_packed_weights()
            params = dict(m.named_parameters())
            rows = {}
            for name in PARITY_TENSORS:
                g, g32 = params[name].grad, leaf[name].grad
                cos = float(torch.nn.functional.cosine_similarity(g.detach().cpu().flatten().double(), g32.flatten().double(), dim=0))
                rows[name.replace("model.", "")] = {"rel_l2": round(rel(g, g32), 4), "cos": round(cos, 4), "finite": bool(torch.isfinite(g).all())}
            l2 = sorted(r["rel_l2"] for r in rows.values())
            return {"semi_rel_l2": round(rel(out["semi"], o["semi"]), 5), "desc_rel_l2": round(rel(out["desc"], o["desc"]), 5),
                    "grad_rel_l2_median": l2[len(l2) // 2], "grad_rel_l2_worst": l2[-1], "grad_cos_min": min(r["cos"] for r in rows.values()), "grads": rows}
        finally:
            net.fp8_train = was
            m.zero_grad(set_to_none=True)
            with torch.no_grad():
                for b, s0 in zip(m.buffers(), bn0):
                    b.copy_(s0)
    rec = {"against": "oracle (PyTorch-CPU fp32) train-mode forward + autograd on the record's final weights, one 640x640 image pair, seeded head projections",
           "bars": "tests/test_gpu_bench_shapes.py: bf16 gradients within 1.15 x the PyTorch-bf16 noise floor (median / p90), cosine > 0.8; f32 compute "
                   "path 2e-3 at this shape (test_config2_gradient_f32_tight_8x640)"}
    if fp8:
        rec["fp8"] = hip_pass(True)
        rec["bf16_same_weights"] = hip_pass(False)
        rec.update({k_: rec["fp8"][k_] for k_ in ("semi_rel_l2", "desc_rel_l2", "grad_rel_l2_median", "grad_rel_l2_worst", "grad_cos_min")})
        rec["note"] = ("8-bit operand rounding is discontinuous: against the fp32 oracle the deep layers decorrelate in ANY fp8 statement of this network "
                       "(tests/test_gpu_fp8.py holds the product to 1.15 x the floor of the fake-quantised oracle, rule for rule)")
    else:
        rec.update(hip_pass(False))
    return rec

TRAIN_GFLOP_PER_SAMPLE = {"n": 27.77, "s": 103.28, "m": 299.67, "l": 657.56}      # 4 F_fwd + 2 F_kp at 640x640


def run_train(a, rank, world, dev, version, batch, steps, warmup, gas=1, size=640, dtype="bf16", parity=True):
    """The reference's optimizer step (src/train.py:189-259) on synthetic batches, data parallel over `world` ranks: `batch` samples per
    GPU and micro-batch, `gas` micro-batches per optimizer step.  Returns the sub-record (rank 0) -- every rank must call it."""
    import torch.distributed as dist
    from yolopoint_amd.utils.synthetic import make_model
    from yolopoint_amd.engine import TrainStep, synthetic_batch
    from yolopoint_amd.dp import timed_region
    fp8 = dtype == "fp8"          # configs[4]: 8-bit Conv operands on top of the bf16 path (engine.TrainStep(fp8=True))
    m, _ = make_model(version, 1234, dtype="bf16" if fp8 else dtype)
    m = m.to(dev).train()
    step = TrainStep(m, dev, img_size=size, gas=gas, fp8=fp8)
    step.comm_events = [] if world > 1 else None
    micro = [synthetic_batch(batch, size, dev, 1234 + rank * 97 + i) for i in range(gas)]
    arg = micro if gas > 1 else micro[0]

    def reduce_max(t):
        if world == 1:
            return t
        x = torch.tensor([t], device=dev)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        return float(x.item())
    wall = timed_region(lambda: step(arg), steps, warmup, torch.cuda.synchronize, (dist.barrier if world > 1 else (lambda: None)), reduce_max)
    exposed, launch_order, rccl_note = None, list(step.reducer.launch_log), None
    if step.comm_events:
        ev = step.comm_events[-steps:]
        exposed = round(sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev), 4)
    if world == 1 and gas == 1 and os.environ.get("YP_BENCH_RCCL_N1", "1") != "0":
        # N = 1: the bucketed all-reduce does not run in the timed region (one rank has nothing to exchange).  So that the RCCL path -- bucket
        # launch order behind the backward plans, async work handles, the compute stream's wait in front of Adam -- is exercised on a single GPU
        # as well, a few EXTRA steps (after the timed region, not part of `value`) run with the collectives forced through a one-rank RCCL group.
        try:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                dist.init_process_group("nccl", rank=0, world_size=1)
            step.reducer.force_collectives, step.comm_events = True, []
            for _ in range(6):
                step(arg)
            torch.cuda.synchronize()
            ev = step.comm_events[2:]
            exposed = round(sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev), 4)
            launch_order = list(step.reducer.launch_log)
            rccl_note = "N = 1: measured on 4 extra steps behind the timed region with the collectives forced through a ONE-rank RCCL group (force_collectives)"
        except Exception as e:          # (a box without a usable RCCL must not lose the record)
            rccl_note = f"one-rank RCCL leg failed: {type(e).__name__}: {e}"[:200]
        finally:
            step.reducer.force_collectives, step.comm_events = False, None
    samples = batch * gas * world * steps
    gflop_sample = TRAIN_GFLOP_PER_SAMPLE.get(version, 0.0) * (size / 640.0) ** 2
    achieved = gflop_sample * samples / wall / 1e3 / max(world, 1)          # TFLOP/s per GPU
    rec = {"workload": f"BASELINE.json configs[{2 if version == 's' else 4}] shape: YOLOPoint-{version} optimizer step as src/train.py:189-259 (pair forward, "
                       f"three losses, native backward, overlapped bucketed all-reduce, one-launch Adam; DESIGN.md 5), {batch} samples/GPU x gas {gas}, {size}x{size}, {dtype}",
           "value": round(2 * samples / wall, 1), "unit": "images/s (an image pair counts as 2 images)", "samples_per_s": round(samples / wall, 1),
           "ms_per_step": round(wall / steps * 1e3, 3), "steps": steps, "warmup": warmup, "n_gpus": world, "dtype": dtype, "scaling": "weak",
           "per_gpu_batch": batch, "gas": gas, "global_batch": batch * gas * world, "parallelism": f"dp{world}",
           "grad_allreduce_bytes": step.reducer.payload_bytes(), "buckets": step.reducer.describe(),
           "bucket_launch_order": launch_order, "exposed_comm_ms_per_step": exposed, "comm_note": rccl_note,
           "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS[dtype], 4),
                        "traffic": train_traffic_record(version, batch, dtype), "per_gpu": True, "algorithmic_gflop_per_sample": round(gflop_sample, 2),
                        "note": "whole step (losses, BN / elementwise passes, optimizer included) against the conv FLOP the reference executes per sample"
                                + ("; priced against the 5 PFLOP/s dense fp8 peak (forward, dgrad AND weight-gradient operands 8-bit where channel counts "
                                   "are multiples of 64; C % 128 layers on block-scaled K = 64 MFMAs)" if fp8 else "")}}
    if world == 1 and rank == 0 and getattr(a, "cpu_threads", None) and not a.no_cpu_baseline and gas == 1:
        rec["cpu_baseline"] = cpu_train_baseline(version, size, a.cpu_threads)
    if world == 1 and rank == 0 and parity and not a.no_cpu_baseline:
        torch.cuda.synchronize()
        step.reducer.force_collectives = False
        try:
            rec["parity"] = train_parity(m, version, dev, fp8, getattr(a, "cpu_threads", None) or min(os.cpu_count() or 1, 16), S=size)
        except Exception as e:          # (a failed check must be visible in the record, not lose the timing)
            rec["parity"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del step, m, micro
    torch.cuda.empty_cache()
    return rec if rank == 0 else None


def bench_train(a, rank, world, dev):
    """--mode train: the training step as the top-level record (per-GPU batch a.batch, a.gas micro-batches per step)."""
    dtype = a.dtype if a.dtype != "f16" else "bf16"
    a.cpu_threads = None if a.no_cpu_baseline else min(os.cpu_count() or 1, 16)
    rec = run_train(a, rank, world, dev, a.version, a.batch, a.steps, a.warmup, gas=a.gas, size=a.size, dtype=dtype)
    if rank != 0:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    top = {"metric": f"images/sec at {a.size}x{a.size} (YOLOPoint-{a.version} training, {a.batch} samples/GPU x gas {a.gas}, {dtype}); an image pair counts as 2 images",
           "value": rec["value"], "unit": "images/s", "samples_per_s": rec["samples_per_s"], "n_gpus": world, "steps": rec["steps"], "warmup": rec["warmup"],
           "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
           "config": {"workload": rec["workload"], "per_gpu_batch": a.batch, "gas": a.gas, "global_batch": rec["global_batch"], "parallelism": rec["parallelism"],
                      "grad_allreduce_bytes": rec["grad_allreduce_bytes"], "buckets": rec["buckets"], "bucket_launch_order": rec["bucket_launch_order"],
                      "exposed_comm_ms_per_step": rec["exposed_comm_ms_per_step"]},
           "roofline": rec["roofline"]}
    if "cpu_baseline" in rec:
        top["cpu_baseline"] = rec["cpu_baseline"]
    emit(json.dumps(top))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def bench_frame(a, dev):
    """--mode frame: the frame pipeline as the top-level record."""
    rec = run_frame(dev, a.version, a.size, a.dtype, a.steps, a.warmup)
    top = {"metric": rec["metric"], "value": rec["value"], "unit": "frames/s", "n_gpus": 1, "steps": rec["steps"], "warmup": rec["warmup"],
           "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": rec["config"], "roofline": rec["roofline"]}
    emit(json.dumps(top))


FWD_GFLOP_PER_IMAGE = {"n": 5.642, "s": 21.023, "m": 61.477, "l": 135.526}       # SURVEY.md 8(d) F_fwd at 640x640


def run_frame(dev, version, S, dtype, steps, warmup, cpu_threads=None):
    """SURVEY.md 8(f) row 2 / BASELINE.json configs[3]: one frame through the GPU-resident front end (forward, keypoint decode +
    NMS, box NMS, box-mask keypoint filter, descriptor sampling) + mutual-NN matching against the previous frame's descriptors."""
    from yolopoint_amd.frontend import YoloPointFrontend
    from yolopoint_amd.models.model_wrap import PointTracker
    from yolopoint_amd.utils.synthetic import synth_image
    m, _ = build_model(version, dtype, dev)
    m.model.use_graph = True          # (two-lane plans replay eagerly on two streams, as the headline record; the keypoint decode / NMS hang into the side lane)
    fe = YoloPointFrontend(m, dev, yolo_config=dict(conf_thres_box=0.25, iou_thres_box=0.45, max_det=300), filter_pts=True)
    frames = [synth_image(1, 3, S, S, 100 + i).to(dev) for i in range(4)]
    # Seeded random heads saturate (every pixel a keypoint, every anchor a box), which is not the load of a trained model: the
    # network runs in full, but the post-processing is fed PLANTED head outputs: a heat map with
    # 1000 x (S/640)^2 Gaussian peaks over U(0, 0.01) noise (as logits of the 65-channel cell softmax) and 2000 box candidates.
    import numpy as np
    from yolopoint_amd.utils.synthetic import planted_heatmap, planted_predictions
    nrows = sum(3 * (S // st) ** 2 for st in (8, 16, 32))
    semis, preds = [], []
    for i in range(len(frames)):
        heat = planted_heatmap(S, S, int(1000 * (S / 640) ** 2), 10 + i).astype(np.float64)
        cells = heat.reshape(S // 8, 8, S // 8, 8).transpose(1, 3, 0, 2).reshape(64, S // 8, S // 8)
        cells = cells / np.maximum(cells.sum(0, keepdims=True), 1.0) * np.minimum(cells.sum(0, keepdims=True), 0.98)
        dust = 1.0 - cells.sum(0, keepdims=True)
        semis.append(torch.from_numpy(np.log(np.concatenate((cells, dust), 0) + 1e-12).astype(np.float32))[None].to(dev))
        preds.append(torch.from_numpy(planted_predictions(1, nrows, 80, 2000, 20 + i, img=S)).to(dev))

    fe.planted = list(zip(semis, preds))          # the real forward, then the planted semi / pred feed the post-processing (the descriptors stay the model's)
    tr = PointTracker()
    prev = [None]
    stats = {}

    def step(i):
        r = fe.process_tensor(frames[i % len(frames)])
        if prev[0] is not None and r["desc"].shape[1] and prev[0].shape[1]:
            stats["matches"] = tr.nn_match_two_way(r["desc"], prev[0], 0.7).shape[1]
        prev[0] = r["desc"]
        stats["keypoints"], stats["boxes"] = int(r["pts"].shape[0]), int(r["boxes"].shape[0])
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gflop = FWD_GFLOP_PER_IMAGE.get(version, 0.0) * (S / 640.0) ** 2
    achieved = gflop * steps / wall / 1e3
    rec = {"metric": f"frames/sec at {S}x{S} (YOLOPoint-{version} frame pipeline: forward + keypoint decode/NMS + box NMS + box-mask filter + descriptor "
                     f"sampling + MNN matching, bs=1, {dtype})",
           "value": round(steps / wall, 1), "unit": "frames/s", "steps": steps, "warmup": warmup, "ms_per_step": round(wall / steps * 1e3, 4), "dtype": dtype,
           "config": {"workload": "BASELINE.json configs[3] shape: one frame end to end, device-resident, 2 host syncs per frame (front-end counters, match count)",
                      "image": [S, S], "post_processing_inputs": "planted heat map / predictions (SURVEY.md 8d), model descriptors", **stats},
           "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS.get(dtype, 2500.0), "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_TFLOPS.get(dtype, 2500.0), 4), "traffic": train_traffic_record("frame_" + version, S, dtype), "algorithmic_gflop_per_frame": round(gflop, 2),
                        "note": "whole frame (post-processing and host syncs included) against the forward's algorithmic conv FLOP"}}
    del fe, m
    torch.cuda.empty_cache()
    if cpu_threads:
        rec["cpu_baseline"] = cpu_frame_baseline(version, S, cpu_threads, [t.cpu() for t in semis], [t.cpu() for t in preds])
    return rec


def bench_export(a, dev):
    """homography-adaptation export, reference configs/coco_export.yaml (100 views per 640x640 image,
    threshold 0.085, nms 4, top_k 1000).  One step = one source image: 100 views through the network as one batch, keypoint
    decode, aggregation of the 100 heat maps in the base frame, threshold + grid NMS, points to the host."""
    import numpy as np
    from yolopoint_amd.export_homography import HomographyExporter
    from yolopoint_amd.utils.synthetic import synth_image
    N, S = 100, a.size
    m, _ = build_model(a.version, a.dtype, dev)
    exp = HomographyExporter(m, dev, dict(nms=4, top_k=1000, detection_threshold=0.085))
    rng = np.random.default_rng(5)
    homs = np.zeros((N, 3, 3), dtype=np.float32)
    for i in range(N):
        ang, sc = rng.uniform(-0.5, 0.5), 1.0 + rng.uniform(-0.25, 0.25)
        homs[i] = [[sc * np.cos(ang), -sc * np.sin(ang), rng.uniform(-0.2, 0.2)], [sc * np.sin(ang), sc * np.cos(ang), rng.uniform(-0.2, 0.2)],
                   [rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15), 1.0]]
    homs[0] = np.eye(3)
    inv = torch.from_numpy(np.linalg.inv(homs.astype(np.float64)).astype(np.float32)).to(dev)
    from yolopoint_amd.utils.loss_functions import warp_image_batch
    base = synth_image(1, 3, S, S, 9).to(dev)
    views = warp_image_batch(base.repeat(N, 1, 1, 1), torch.from_numpy(homs).to(dev), device=dev).contiguous()
    mask = warp_image_batch(torch.ones(N, 1, S, S, device=dev), torch.from_numpy(homs).to(dev), device=dev, mode="nearest").contiguous()
    sample = {"image": views[None], "valid_mask": mask.view(1, N, S, S), "inv_homographies": inv[None]}
    # random-weight heads give a flat heat map (no point reaches 0.085): the network runs in full, the aggregation / decode is fed
    # planted keypoint logits (1000 x (S/640)^2 peaks in the base frame, SURVEY.md 8d, warped into every view), as in --mode frame
    from yolopoint_amd.utils.synthetic import planted_heatmap
    heat = torch.from_numpy(planted_heatmap(S, S, int(1000 * (S / 640) ** 2), 10).astype(np.float32)).to(dev)
    hv = warp_image_batch(heat[None, None].repeat(N, 1, 1, 1), torch.from_numpy(homs).to(dev), device=dev)          # the peaks as each view sees them
    cells = torch.nn.functional.pixel_unshuffle(hv, 8).double()                                                       # [N,64,S/8,S/8]
    tot = cells.sum(1, keepdim=True)
    cells = cells / tot.clamp_min(1.0) * tot.clamp_max(0.98)
    semi = torch.log(torch.cat((cells, 1.0 - cells.sum(1, keepdim=True)), 1) + 1e-12).float().contiguous()

    class PlantedSemi(torch.nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

        def forward(self, x):
            self.model(x)
            return {"semi": semi}
    exp.model = PlantedSemi(m)
    npts = 0
    for _ in range(max(1, a.warmup // 5)):
        npts = exp.export_sample(sample).shape[0]
    torch.cuda.synchronize()
    steps = max(1, a.steps // 5)
    t0 = time.perf_counter()
    for _ in range(steps):
        npts = exp.export_sample(sample).shape[0]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    emit(json.dumps({"metric": f"source images/sec, homography-adaptation export ({N} views of {S}x{S} per image, YOLOPoint-{a.version}, {a.dtype})",
                      "value": round(steps / wall, 2), "unit": "images/s", "views_per_s": round(steps * N / wall, 1), "n_gpus": 1, "steps": steps,
                      "warmup": max(1, a.warmup // 5), "ms_per_step": round(wall / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                      "config": {"workload": "reference configs/coco_export.yaml: 100 views, detection_threshold 0.085, nms 4, top_k 1000", "points": npts}}))


def bench_postproc(dev):
    """Post-processing kernels on planted head outputs; microseconds per call, GPU events."""
    import numpy as np
    from yolopoint_amd.utils.synthetic import planted_heatmap, planted_predictions, planted_descriptors
    from yolopoint_amd.utils import utils as U
    from yolopoint_amd.utils.general_yolo import non_max_suppression
    from yolopoint_amd.models.model_wrap import PointTracker
    res = {}

    def timeit(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / n * 1e6, 1)

    semi = torch.randn(8, 65, 80, 80, device=dev)
    res["kp_decode_b8_us"] = timeit(lambda: U.flattenDetection(semi))
    heat = torch.from_numpy(planted_heatmap(640, 640, 1000, 1)).to(dev)
    res["kp_nms_640_1000peaks_us"] = timeit(lambda: U.getPtsFromHeatmap(heat, 0.015, 4))
    pred = torch.from_numpy(planted_predictions(8, 25200, 80, 2000, 1)).to(dev)
    res["box_nms_b8_25200x85_2000cand_us"] = timeit(lambda: non_max_suppression(pred, 0.25, 0.45, labels=[], multi_label=True, agnostic=True, max_det=1000), 5)
    d1, d2 = planted_descriptors(256, 1000, 1000, 0.7, 1)
    d1, d2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    tr = PointTracker()
    res["mnn_256d_1000x1000_us"] = timeit(lambda: tr.nn_match_two_way(d1, d2, 0.7))
    return res


_REAL_STDOUT_FD = None


def emit(line):
    """The ONE JSON line of the contract.  Libraries write to the process's stdout as well (RCCL prints a version banner from C stdio when its
    first communicator comes up -- also for the one-rank group of the N = 1 `train` record -- and libc flushes it at exit, BEHIND anything Python
    printed): the whole run therefore has file descriptor 1 pointed at stderr, and only this line goes to the real stdout."""
    if _REAL_STDOUT_FD is None:
        print(line, flush=True)
        return
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.write(_REAL_STDOUT_FD, (line + "\n").encode())


if __name__ == "__main__":
    sys.stdout.flush()
    _args = parse()
    if "WORLD_SIZE" not in os.environ and _args.gpus > 1:
        self_launch(_args, sys.argv[1:])          # exec: does not return (before file descriptor 1 is redirected below)
    _REAL_STDOUT_FD = os.dup(1)
    os.dup2(2, 1)           # everything else any library prints: stderr
    main(_args)
