#!/usr/bin/env python3
"""Distil the rocprofv3 CSVs written by tools/profile_round.sh into gpurun_out/prof_<round>/summary files
(copied into profiles/ by hand after review).  Plan executions per bench run = steps + warmup + 5 profile passes."""
import collections, csv, glob, json, os, re, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
CONV = re.compile(r"conv_igemm_kernel|conv_mma8_kernel|conv_wsk_kernel|conv3x3_halo_kernel|bottleneck_halo_kernel|bneck32_persist_kernel|stem_conv_kernel|stem_conv2_kernel")
STEM = re.compile(r"stem_conv_kernel|stem_conv2_kernel")
STEPS, WARMUP = 50, 10


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:100]


def find(sub, pat):
    fs = glob.glob(os.path.join(OUT, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def expected_launches():
    """(conv launches, all launches) of one plan execution as bench.py itself counts them (roofline.launches_per_step / config.ops_per_step of the
    un-profiled run of the same command, bench_long.json)."""
    try:
        d = json.loads(open(os.path.join(OUT, "bench_long.json")).read().strip().splitlines()[-1])
        return int(d["roofline"]["launches_per_step"]), int(d["config"]["ops_per_step"])
    except Exception:
        return None, None


def plan_windows(path):
    """The plan executions of the TIMED region of `bench.py --only none`: a forward starts with the stem launch, so the dispatches between two
    consecutive stem launches (in dispatch order = host enqueue order; the two lanes of a forward are enqueued before the next forward's stem) are
    one execution.  Plan construction (the autotuner launches every candidate of every convolution), the five per-launch profile passes and the
    stand-alone stem timings that bench.py runs in the same process give windows of other shapes: a window counts only if its kernel-name
    sequence is the modal one.  Returns (list of windows, each a list of csv rows; the modal signature's length)."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if STEM.search(r["Kernel_Name"])]
    wins = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    sig = collections.Counter(tuple(short(r["Kernel_Name"]) for r in w) for w in wins)
    if not sig:
        raise SystemExit(f"profile_collect: no stem launch in {path}: the marker regex no longer matches the plan's first kernel")
    modal, n = sig.most_common(1)[0]
    good = [w for w in wins if tuple(short(r["Kernel_Name"]) for r in w) == modal]
    return good, len(modal)


def check_windows(wins, nk, what):
    n_conv_exp, n_all_exp = expected_launches()
    n_conv = sum(1 for r in wins[0] if CONV.search(r["Kernel_Name"]))
    if len(wins) < STEPS:
        raise SystemExit(f"profile_collect[{what}]: only {len(wins)} plan executions of the modal shape, expected >= {STEPS} timed steps")
    if n_conv_exp is not None and (n_conv != n_conv_exp or nk != n_all_exp):
        raise SystemExit(f"profile_collect[{what}]: a plan execution has {n_conv} convolution launches / {nk} launches, bench.py counts {n_conv_exp} / {n_all_exp}: "
                         "the CONV / STEM regexes are out of date or the window is polluted")
    return n_conv


def trace_table(wins, title):
    d = collections.defaultdict(list)
    for w in wins:
        for r in w:
            d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    lines = [f"# {title}", f"{'kernel':102s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}"]
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k:102s} {len(v):7d} {sum(v)/1e3:11.1f} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:8.2f} {max(v)/1e3:8.2f} {100*sum(v)/tot:6.2f}")
    return d, lines


def counter_windows(sub, counter):
    """Per plan execution: [(kernel name, counter value)] in dispatch order, from the counter CSV of a --pmc pass (windows as plan_windows)."""
    path = find(sub, "*counter_collection.csv")
    agg = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])                       # (a counter may be reported as several rows per dispatch: one per instance)
        if k in agg:
            agg[k]["Counter_Value"] = float(agg[k]["Counter_Value"]) + float(r["Counter_Value"])
        else:
            agg[k] = dict(r)
    rows = [agg[k] for k in sorted(agg)]
    starts = [i for i, r in enumerate(rows) if STEM.search(r["Kernel_Name"])]
    wins = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    sig = collections.Counter(tuple(short(r["Kernel_Name"]) for r in w) for w in wins)
    if not sig:
        raise SystemExit(f"profile_collect: no stem launch in {path}")
    modal, _ = sig.most_common(1)[0]
    good = [w for w in wins if tuple(short(r["Kernel_Name"]) for r in w) == modal]
    return good, len(modal)


def layer_names(fname):
    path = os.path.join(OUT, fname)
    if not os.path.exists(path):
        return []
    return [l.split()[:2] for l in open(path) if l.strip() and not l.startswith("#") and not l.startswith("op ")]


def main():
    cmd = f"python bench.py --no-cpu-baseline --only none --steps {STEPS} --warmup {WARMUP}"
    res = {"round": R, "command": cmd, "window": "dispatches between consecutive stem launches whose kernel sequence is the modal one (= the plan executions of "
                                                  "the warm-up and timed loops; plan construction, profile passes and stand-alone stem timings are other shapes)"}
    lines = []
    t = find("trace", "*kernel_trace.csv")
    if t:
        wins, nk = plan_windows(t)
        n_conv = check_windows(wins, nk, "trace")
        d, tl = trace_table(wins, f"rocprofv3 --kernel-trace of: {cmd}  ({len(wins)} plan executions of {nk} launches each)")
        lines += tl
        conv_ns = sum(sum(v) for k, v in d.items() if CONV.search(k))
        spans = sorted(max(int(r["End_Timestamp"]) for r in w) - int(w[0]["Start_Timestamp"]) for w in wins)
        res.update(plan_executions=len(wins), launches_per_step=nk, conv_launches_per_step=n_conv, conv_us_per_step_trace=conv_ns / 1e3 / len(wins),
                   conv_avg_us_per_launch_trace=conv_ns / 1e3 / (n_conv * len(wins)), all_kernels_us_per_step_trace=sum(sum(v) for v in d.values()) / 1e3 / len(wins),
                   step_span_us_median=spans[len(spans) // 2] / 1e3)
    if find("pmc_fetch", "*counter_collection.csv") and find("pmc_write", "*counter_collection.csv"):
        fw, nkf = counter_windows("pmc_fetch", "FETCH_SIZE")
        ww, nkw = counter_windows("pmc_write", "WRITE_SIZE")
        check_windows(fw, nkf, "pmc_fetch"); n_conv = check_windows(ww, nkw, "pmc_write")
        # rocprofv3 reports both counters in KiB (check: l2norm writes 51200 px x 128 ch x 4 B = 25 600 KiB per launch, the value it
        # shows); gfx950 FETCH_SIZE counts 16-B/lane streaming reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> x2.
        conv_sum = lambda wins: sum(float(r["Counter_Value"]) for w in wins for r in w if CONV.search(r["Kernel_Name"])) * 1024 / len(wins)
        all_sum = lambda wins: sum(float(r["Counter_Value"]) for w in wins for r in w) * 1024 / len(wins)
        fetch_b, write_b = conv_sum(fw), conv_sum(ww)
        traffic = {"source": f"gpurun_out/prof_{R}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `{cmd}`",
                   "workload": "YOLOPoint-s bs8 640x640 f16", "plan_executions": [len(fw), len(ww)], "window": res["window"],
                   "kernels": "the convolution launches of the plan (stem_conv2 / conv_igemm / conv_mma8 / conv3x3_halo / bottleneck_halo / chain kernels, all instantiations)",
                   "conv_launches_per_step": n_conv,
                   "fetch_bytes_per_step_raw": fetch_b, "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests of 16-B/lane streaming reads at 64 B)",
                   "write_bytes_per_step_raw": write_b, "hbm_bytes_per_step": 2 * fetch_b + write_b,
                   "hbm_bytes_per_launch": (2 * fetch_b + write_b) / n_conv,
                   "all_kernels_hbm_bytes_per_step": 2 * all_sum(fw) + all_sum(ww),
                   "conv_us_per_step_trace": res.get("conv_us_per_step_trace")}
        json.dump(traffic, open(os.path.join(OUT, "conv_traffic.json"), "w"), indent=1)
        res["traffic"] = traffic
    # per launch (one-lane eager plan: dispatch order = plan order = the rows of layers_1lane.txt): counter bytes beside the algorithmic bytes
    if find("pmc_fetch1", "*counter_collection.csv") and find("pmc_write1", "*counter_collection.csv"):
        fw, nkf = counter_windows("pmc_fetch1", "FETCH_SIZE")
        ww, nkw = counter_windows("pmc_write1", "WRITE_SIZE")
        names = layer_names("layers_1lane.txt")
        tw, _ = plan_windows(find("pmc_fetch1", "*kernel_trace.csv"))
        if names and len(names) == nkf == nkw:
            tbl = [f"# per launch, YOLOPoint-s bs8 640x640 f16, ONE-LANE eager plan (YP_INFER_LANES=0 --no-graph: dispatch order = plan order) under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE",
                   f"# (separate passes; mean over {len(fw)} / {len(ww)} plan executions; FETCH_SIZE x2, KiB -> bytes; alg = the layer's input + output + filter bytes of layers_1lane.txt)",
                   f"{'op':52s} {'kernel':28s} {'us(pmc run)':>11s} {'fetch_MB':>9s} {'write_MB':>9s} {'hbm_MB':>8s} {'alg_MB':>8s} {'ratio':>6s}"]
            alg = {}
            for l in open(os.path.join(OUT, "layers_1lane.txt")):
                p_ = l.split()
                if len(p_) >= 8 and not l.startswith("#") and p_[0] != "op":
                    alg[p_[0]] = float(p_[7]) * 1e9 * float(p_[5]) * 1e-6          # GB/s(alg) x us
            for i, (nm, kind) in enumerate(names):
                fb = 2 * 1024 * sum(float(w[i]["Counter_Value"]) for w in fw) / len(fw)
                wb = 1024 * sum(float(w[i]["Counter_Value"]) for w in ww) / len(ww)
                us = sum(int(w[i]["End_Timestamp"]) - int(w[i]["Start_Timestamp"]) for w in tw) / len(tw) / 1e3
                a_ = alg.get(nm, 0.0)
                tbl.append(f"{nm:52s} {short(fw[0][i]['Kernel_Name'])[:28]:28s} {us:11.1f} {fb / 1e6:9.2f} {wb / 1e6:9.2f} {(fb + wb) / 1e6:8.2f} {a_ / 1e6:8.2f} {((fb + wb) / a_ if a_ else 0):6.2f}")
            open(os.path.join(OUT, f"{R}_layers_traffic.txt"), "w").write("\n".join(tbl) + "\n")
        else:
            print(f"profile_collect: per-launch traffic table skipped ({len(names)} layer rows, {nkf} / {nkw} dispatches per execution)", file=sys.stderr)
    mf = find("pmc_mfma", "*counter_collection.csv")
    if mf:
        # per dispatch: MFMA-busy cycles summed over the chip's SIMDs / (GPU-active cycles x 4 SIMDs x 256 CUs).  One-lane eager replay: the k-th
        # kernel of a plan execution is the k-th op of the plan (stem first), so the backbone (Conv1..SPPooling) can be separated.
        bw, nkb = counter_windows("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
        gw, nkg = counter_windows("pmc_mfma", "GRBM_GUI_ACTIVE")
        names = [n for n, _ in layer_names("layers_1lane.txt")]
        bbn = ("Conv1", "Conv2", "Bottleneck1", "Conv3", "Bottleneck2", "Conv4", "Bottleneck3", "Conv5", "Bottleneck4", "SPPooling")
        tot = {"all": [0.0, 0.0], "conv": [0.0, 0.0], "backbone": [0.0, 0.0]}
        for wb_, wg_ in zip(bw, gw):
            for i, (rb, rg) in enumerate(zip(wb_, wg_)):
                busy, act = float(rb["Counter_Value"]), float(rg["Counter_Value"])
                tot["all"][0] += busy; tot["all"][1] += act
                if CONV.search(rb["Kernel_Name"]):
                    tot["conv"][0] += busy; tot["conv"][1] += act
                    if len(names) == nkb and names[i].split(".")[0] in bbn:
                        tot["backbone"][0] += busy; tot["backbone"][1] += act
        # Calibration (tools/probe/pmc_calib.sh, same pass = counters + kernel trace): SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of
        # their MFMA-busy cycles -- exactly 16 cycles x the number of v_mfma_f32_16x16x32 instructions (102 400 MFMAs of a 256->256 1x1
        # layer at M = 12 800 -> 1 638 400) -- and GRBM_GUI_ACTIVE is reported summed over 16 counter instances on this stack (value /
        # dispatch duration = 30-35 per ns at a ~2.1 GHz clock).  Utilisation = busy SIMD-cycles / (GPU-active cycles x 1024 SIMDs).
        GUI_INSTANCES = 16
        util = {k: (v[0] / (v[1] / GUI_INSTANCES * 4 * 256) if v[1] else None) for k, v in tot.items()}
        rec = {"source": f"gpurun_out/prof_{R}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass, --kernel-trace only) of `YP_INFER_LANES=0 {cmd} --no-graph`",
               "formula": "sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 16 instances x 4 SIMDs x 256 CUs) over the dispatches of the plan executions "
                          "(windows between stem launches, modal kernel sequence); calibrated: MFMA_BUSY = 16 cycles x (number of 16x16x32 MFMAs), GUI_ACTIVE / duration = 16 x clock",
               "workload": "YOLOPoint-s bs8 640x640 f16", "plan_executions": len(bw), "launches_per_execution": nkb,
               "mfma_busy_all_kernels": util["all"], "mfma_busy_conv_kernels": util["conv"], "mfma_busy_backbone_convs": util["backbone"]}
        json.dump(rec, open(os.path.join(OUT, "mfma_busy.json"), "w"), indent=1)
        res["mfma_busy"] = rec
    open(os.path.join(OUT, f"{R}_infer_kernel_trace.txt"), "w").write("\n".join(lines) + "\n")
    tt = find("trace_train", "*kernel_trace.csv")
    if tt:
        d, tl = trace_table([list(csv.DictReader(open(tt)))], "rocprofv3 --kernel-trace of: python bench.py --mode train --steps 5 --warmup 2 (7 optimizer steps + plan build/autotune)")
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
