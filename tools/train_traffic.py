#!/usr/bin/env python3
"""HBM bytes of ONE optimizer step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`python bench.py --mode train ...`: the dispatches between the last two adam_flat_kernel launches.  FETCH_SIZE x2 (gfx950 tallies the
128-byte requests of 16-B/lane streaming reads at 64 B: MI355X_MICROARCH.md, HBM section), both counters in KiB.
usage: tools/train_traffic.py <fetch_dir> <write_dir> <key> <out.json> [marker kernel, default adam_flat_kernel]
(the frame pipeline: marker mnn_select_kernel, one per frame)"""
import csv, glob, json, sys


MARKER = sys.argv[5] if len(sys.argv) > 5 else "adam_flat_kernel"


def step_sum(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adam = [i for i, r in enumerate(rows) if MARKER in r["Kernel_Name"]]
    win = rows[adam[-2] + 1:adam[-1] + 1]
    return sum(float(r["Counter_Value"]) for r in win) * 1024.0, len(win)


fetch, n1 = step_sum(sys.argv[1], "FETCH_SIZE")
write, n2 = step_sum(sys.argv[2], "WRITE_SIZE")
key, out = sys.argv[3], sys.argv[4]
try:
    d = json.load(open(out))
except Exception:
    d = {}
d[key] = {"hbm_bytes_per_step": round(2 * fetch + write), "fetch_bytes_raw": round(fetch), "write_bytes_raw": round(write), "kernels_in_step": n1,
          "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --mode train | frame; FETCH_SIZE x2 (gfx950), KiB -> bytes; "
                    f"dispatches between the last two {MARKER} launches"}
json.dump(d, open(out, "w"), indent=1)
print(key, d[key]["hbm_bytes_per_step"] / 1e9, "GB per step,", n1, "kernels")
