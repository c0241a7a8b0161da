#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection) into a small text + JSON file.

  python tools/rocprof_summary.py --trace X_kernel_trace.csv [--pmc A_counter_collection.csv ...] --steps N --out profiles/NAME

Per kernel: launches, total/avg/min/max duration; for PMC files: per-kernel sums of each counter.  `--steps` is the
number of bench steps (timed + warm-up + profile passes) so per-step figures can be derived."""
import argparse, collections, csv, json, re

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace"); ap.add_argument("--pmc", nargs="*", default=[]); ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    res = {"note": a.note, "kernels": {}, "counters": {}}
    lines = []
    if a.trace:
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(a.trace)):
            d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v) for v in d.values())
        lines.append(f"{'kernel':92s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            res["kernels"][k] = {"calls": len(v), "total_us": sum(v) / 1e3, "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
            lines.append(f"{k:92s} {len(v):7d} {sum(v)/1e3:11.1f} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:8.2f} {max(v)/1e3:8.2f} {100*sum(v)/tot:6.2f}")
    for f in a.pmc:
        d = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(lambda: collections.defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]
            d[k][c] += float(r["Counter_Value"]); n[k][c] += 1
        for k in d:
            for c in d[k]:
                res["counters"].setdefault(k, {})[c] = {"sum": d[k][c], "dispatches": n[k][c]}
        lines.append("")
        lines.append(f"# counters from {f.split('/')[-1]} (sum over dispatches)")
        for k in sorted(d, key=lambda k: -max(d[k].values())):
            lines.append(f"{k:92s} " + "  ".join(f"{c}={d[k][c]:.4g}/{n[k][c]}" for c in sorted(d[k])))
    open(a.out + ".txt", "w").write((("# " + a.note + "\n") if a.note else "") + "\n".join(lines) + "\n")
    json.dump(res, open(a.out + ".json", "w"), indent=1)
    print("\n".join(lines[:14]))

if __name__ == "__main__":
    main()
