#!/usr/bin/env python3
"""Distil the rocprofv3 CSVs written by tools/profile_round.sh into gpurun_out/prof_<round>/summary files
(copied into profiles/ by hand after review).  Plan executions per bench run = steps + warmup + 5 profile passes."""
import collections, csv, glob, json, os, re, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
CONV = re.compile(r"conv_igemm_kernel|conv_mma8_kernel|conv3x3_halo_kernel|bottleneck_halo_kernel|stem_conv_kernel")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:100]


def find(sub, pat):
    fs = glob.glob(os.path.join(OUT, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def first_step_dispatch(path):
    """Dispatch id of the first stem_conv_kernel launch: everything before it is plan construction (kernel-variant
    autotuning launches every candidate of every convolution), everything from it on is the measured plan replays."""
    ids = [int(r["Dispatch_Id"]) for r in csv.DictReader(open(path)) if "stem_conv_kernel" in r["Kernel_Name"]]
    return min(ids) if ids else 0


def trace_table(path, execs, title, steady=True):
    d = collections.defaultdict(list)
    d0 = first_step_dispatch(path) if steady else 0
    for r in csv.DictReader(open(path)):
        if int(r["Dispatch_Id"]) < d0:
            continue
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    lines = [f"# {title}", f"{'kernel':102s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}"]
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k:102s} {len(v):7d} {sum(v)/1e3:11.1f} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:8.2f} {max(v)/1e3:8.2f} {100*sum(v)/tot:6.2f}")
    return d, lines


def counter_sum(path, counter, d0):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if int(r["Dispatch_Id"]) < d0:
            continue
        if r["Counter_Name"] == counter and CONV.search(r["Kernel_Name"]):
            s += float(r["Counter_Value"]); n += 1
    return s, n


def main():
    execs = 50 + 10 + 5
    res = {"round": R, "command": "python bench.py --no-cpu-baseline --steps 50 --warmup 10", "plan_executions": execs}
    lines = []
    t = find("trace", "*kernel_trace.csv")
    if t:
        d, tl = trace_table(t, execs, "rocprofv3 --kernel-trace of: " + res["command"] + "  (dispatches from the first plan replay on: plan construction / autotuning excluded)")
        lines += tl
        conv_ns = sum(sum(v) for k, v in d.items() if CONV.search(k))
        conv_calls = sum(len(v) for k, v in d.items() if CONV.search(k))
        res.update(conv_launches_per_step=conv_calls / execs, conv_us_per_step_trace=conv_ns / 1e3 / execs,
                   conv_avg_us_per_launch_trace=conv_ns / 1e3 / max(conv_calls, 1),
                   all_kernels_us_per_step_trace=sum(sum(v) for v in d.values()) / 1e3 / execs)
    f, w = find("pmc_fetch", "*counter_collection.csv"), find("pmc_write", "*counter_collection.csv")
    if f and w:
        fs, fn = counter_sum(f, "FETCH_SIZE", first_step_dispatch(find("pmc_fetch", "*kernel_trace.csv")))
        ws, wn = counter_sum(w, "WRITE_SIZE", first_step_dispatch(find("pmc_write", "*kernel_trace.csv")))
        # rocprofv3 reports both counters in KiB (check: l2norm writes 51200 px x 128 ch x 4 B = 25 600 KiB per launch, the value it
        # shows); gfx950 FETCH_SIZE counts 16-B/lane streaming reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> x2.
        fetch_b, write_b = fs * 1024 / execs, ws * 1024 / execs
        traffic = {"source": f"gpurun_out/prof_{R}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `{res['command']}`",
                   "workload": "YOLOPoint-s bs8 640x640 f16", "plan_executions": execs,
                   "kernels": "conv_igemm_kernel / conv3x3_halo_kernel / bottleneck_halo_kernel / stem_conv_kernel (all instantiations)",
                   "conv_launches_per_step": res.get("conv_launches_per_step"),
                   "fetch_bytes_per_step_raw": fetch_b, "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests of 16-B/lane streaming reads at 64 B)",
                   "write_bytes_per_step_raw": write_b, "hbm_bytes_per_step": 2 * fetch_b + write_b,
                   "hbm_bytes_per_launch": (2 * fetch_b + write_b) / max(res.get("conv_launches_per_step") or 1, 1),
                   "conv_us_per_step_trace": res.get("conv_us_per_step_trace")}
        json.dump(traffic, open(os.path.join(OUT, "conv_traffic.json"), "w"), indent=1)
        res["traffic"] = traffic
    mf = find("pmc_mfma", "*counter_collection.csv")
    if mf:
        # per dispatch: MFMA-busy cycles summed over the chip's SIMDs / (GPU-active cycles x 4 SIMDs x 256 CUs).  Eager replay: the k-th
        # kernel of a plan execution is the k-th op of the plan (stem first), so the backbone (Conv1..SPPooling) can be separated.
        d0 = first_step_dispatch(find("pmc_mfma", "*kernel_trace.csv"))
        rows = collections.defaultdict(dict)
        for r in csv.DictReader(open(mf)):
            if int(r["Dispatch_Id"]) >= d0:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                rows[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
        order = [rows[k] for k in sorted(rows)]
        layers = [l.split()[0] for l in open(os.path.join(OUT, "layers.txt")) if l.strip() and not l.startswith("#") and not l.startswith("op ")] if os.path.exists(os.path.join(OUT, "layers.txt")) else []
        bbn = ("Conv1", "Conv2", "Bottleneck1", "Conv3", "Bottleneck2", "Conv4", "Bottleneck3", "Conv5", "Bottleneck4", "SPPooling")
        per = len(layers)
        tot = {"all": [0.0, 0.0], "conv": [0.0, 0.0], "backbone": [0.0, 0.0]}
        for i, r in enumerate(order):
            busy, act = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), r.get("GRBM_GUI_ACTIVE", 0.0)
            tot["all"][0] += busy; tot["all"][1] += act
            if CONV.search(r["name"]):
                tot["conv"][0] += busy; tot["conv"][1] += act
                if per and layers[i % per].split(".")[0] in bbn:
                    tot["backbone"][0] += busy; tot["backbone"][1] += act
        # Calibration (tools/probe/pmc_calib.sh, same pass = counters + kernel trace): SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of
        # their MFMA-busy cycles -- exactly 16 cycles x the number of v_mfma_f32_16x16x32 instructions (102 400 MFMAs of a 256->256 1x1
        # layer at M = 12 800 -> 1 638 400) -- and GRBM_GUI_ACTIVE is reported summed over 16 counter instances on this stack (value /
        # dispatch duration = 30-35 per ns at a ~2.1 GHz clock).  Utilisation = busy SIMD-cycles / (GPU-active cycles x 1024 SIMDs).
        GUI_INSTANCES = 16
        util = {k: (v[0] / (v[1] / GUI_INSTANCES * 4 * 256) if v[1] else None) for k, v in tot.items()}
        rec = {"source": f"gpurun_out/prof_{R}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass, --kernel-trace only) of `{res['command']} --no-graph`",
               "formula": "sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 16 instances x 4 SIMDs x 256 CUs) over the dispatches of the plan replays; "
                          "calibrated: MFMA_BUSY = 16 cycles x (number of 16x16x32 MFMAs), GUI_ACTIVE / duration = 16 x clock",
               "workload": "YOLOPoint-s bs8 640x640 f16", "dispatches": len(order),
               "mfma_busy_all_kernels": util["all"], "mfma_busy_conv_kernels": util["conv"], "mfma_busy_backbone_convs": util["backbone"]}
        json.dump(rec, open(os.path.join(OUT, "mfma_busy.json"), "w"), indent=1)
        res["mfma_busy"] = rec
    open(os.path.join(OUT, f"{R}_infer_kernel_trace.txt"), "w").write("\n".join(lines) + "\n")
    tt = find("trace_train", "*kernel_trace.csv")
    if tt:
        d, tl = trace_table(tt, 7, "rocprofv3 --kernel-trace of: python bench.py --mode train --steps 5 --warmup 2 (7 optimizer steps + plan build/autotune)", steady=False)
        open(os.path.join(OUT, f"{R}_train_kernel_trace.txt"), "w").write("\n".join(tl[:80]) + "\n")
        # one steady-state optimizer step: the dispatches between the last two Adam kernels
        rows = sorted(csv.DictReader(open(tt)), key=lambda r: int(r["Start_Timestamp"]))
        adam = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"] or "fused_adam" in r["Kernel_Name"].lower() or "adam_flat_kernel" in r["Kernel_Name"]]
        ends = [i for i in adam if i + 1 >= len(rows) or (i + 1) not in set(adam)]       # last kernel of each step's optimizer run
        if len(ends) >= 2:
            win = rows[ends[-2] + 1:ends[-1] + 1]
            agg = collections.defaultdict(list)
            for r in win:
                agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            tot = sum(sum(v) for v in agg.values())
            span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
            sl = [f"# ONE steady-state training step (dispatches between the last two Adam kernels): {len(win)} kernels, busy {tot / 1e3:.0f} us, "
                  f"span {span / 1e3:.0f} us", f"{'kernel':102s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'%':>6s}"]
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                sl.append(f"{k:102s} {len(v):7d} {sum(v) / 1e3:11.1f} {sum(v) / len(v) / 1e3:9.2f} {100 * sum(v) / tot:6.2f}")
            open(os.path.join(OUT, f"{R}_train_step_kernels.txt"), "w").write("\n".join(sl[:70]) + "\n")
            # the same step in launch order: start offset, duration, gap to the previous kernel's end
            t0, prev = int(win[0]["Start_Timestamp"]), None
            seq = ["# launch order of the step above: start_us dur_us gap_us(to the previous kernel's end, any lane) lane kernel"]
            lane_key = "Stream_Id" if "Stream_Id" in win[0] and len({r["Stream_Id"] for r in win}) > 1 else ("Queue_Id" if "Queue_Id" in win[0] else None)
            lanes = {}
            for r in win:
                st_, en_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                ln = r[lane_key] if lane_key else "0"
                lanes.setdefault(ln, []).append((st_, en_, short(r["Kernel_Name"])[:60]))
                seq.append(f"{(st_ - t0) / 1e3:9.1f} {(en_ - st_) / 1e3:8.1f} {((st_ - prev) / 1e3 if prev else 0):7.1f}  L{ln:<3s} {short(r['Kernel_Name'])[:110]}")
                prev = en_
            # per lane (HIP stream / hardware queue): busy time, and the waits of the lane with the most kernels (the step's main chain) --
            # every pause of more than 10 us between two of ITS kernels, with what the other lanes ran meanwhile
            seq.append(f"# lanes by {lane_key}: " + "; ".join(f"L{k}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e3:.0f} us, "
                                                               f"from {(v[0][0] - t0) / 1e3:.0f} to {(v[-1][1] - t0) / 1e3:.0f} us" for k, v in lanes.items()))
            main = max(lanes, key=lambda k: len(lanes[k]))
            mv = lanes[main]
            waits = [(mv[i + 1][0] - mv[i][1], mv[i][1], mv[i + 1][0], mv[i][2], mv[i + 1][2]) for i in range(len(mv) - 1) if mv[i + 1][0] - mv[i][1] > 10000]
            seq.append(f"# main lane L{main}: {len(waits)} pauses > 10 us between its own kernels, {sum(w[0] for w in waits) / 1e3:.0f} us in all")
            for w, a_, b_, ka, kb in waits:
                other = sum(min(e, b_) - max(s_, a_) for k, v in lanes.items() if k != main for s_, e, _ in v if e > a_ and s_ < b_)
                seq.append(f"#   {w / 1e3:7.1f} us at {(a_ - t0) / 1e3:8.1f}: {ka} -> {kb}; other lanes busy {other / 1e3:.1f} us meanwhile")
            open(os.path.join(OUT, f"{R}_train_step_sequence.txt"), "w").write("\n".join(seq) + "\n")
    json.dump(res, open(os.path.join(OUT, f"{R}_collect.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
