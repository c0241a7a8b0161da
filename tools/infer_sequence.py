#!/usr/bin/env python3
"""One graph replay of the inference step from a rocprofv3 kernel-trace CSV, in start order: start offset, duration, gap to the end of the
latest-ending earlier kernel (negative = overlap), kernel name.  python tools/infer_sequence.py trace_kernel_trace.csv [ops_per_step]"""
import csv, re, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 51
# the last complete step = the last n kernels that start with the stem kernel
idx = [i for i, r in enumerate(rows) if "stem_conv" in r[2]]
cand = [(a_, b_) for a_, b_ in zip(idx[:-1], idx[1:]) if n - 2 <= b_ - a_ <= n + 2]      # (one stem launch + the plan's launches)
i0, i1 = cand[len(cand) // 2]
step = rows[i0:i1]
t0 = step[0][0]
def short(nm):
    nm = re.sub(r"\(anonymous namespace\)::", "", nm); nm = re.sub(r"^void ", "", nm)
    return nm.split("(")[0][:70]
latest = None
print(f"# {len(step)} kernels, span {(max(r[1] for r in step) - t0) / 1e3:.1f} us, busy sum {sum(r[1] - r[0] for r in step) / 1e3:.1f} us")
print(f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}  kernel")
for s, e, nm in step:
    gap = (s - latest) / 1e3 if latest is not None else 0.0
    print(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:8.2f}  {short(nm)}")
    latest = e if latest is None else max(latest, e)
