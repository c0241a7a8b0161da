#!/usr/bin/env python3
"""Three-parameter cost model of the inference plan's convolution launches, fitted to a per-launch table (profiles/rNN_layers_infer.txt):
    us = a + b * GFLOP + c * MB      (least squares over the conv launches)
a = what a launch costs whatever it does, 1/b = the marginal MFMA rate, 1/c = the marginal HBM rate.  Usage: tools/layer_model.py [table]"""
import re
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02_layers_infer.txt"
rows = []
for line in open(path):
    m = re.match(r"(\S+)\s+conv\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)", line)
    if m:
        name, M, N, K, us, tf, gbs = m.groups()
        us = float(us)
        rows.append((name, 2 * int(M) * int(N) * int(K) / 1e9, float(gbs) * us / 1e3, us))
A = np.array([[1.0, r[1], r[2]] for r in rows])
y = np.array([r[3] for r in rows])
(a, b, c), *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ np.array([a, b, c])
print(f"{len(rows)} conv launches, {y.sum():.0f} us measured")
print(f"us = {a:.2f} + {b:.3f} * GFLOP + {c:.4f} * MB     (rms error {np.sqrt(np.mean((pred - y) ** 2)):.1f} us)")
print(f"per-launch constant: {a * len(rows):.0f} us of the total; FLOP term {A[:, 1].sum() * b:.0f} us = {1 / b:.2f} PFLOP/s marginal; "
      f"byte term {A[:, 2].sum() * c:.0f} us = {1 / c:.1f} TB/s marginal")
for n, fl, mb, us in rows:
    print(f"  {n:44s} {us:7.1f} us  model {a + b * fl + c * mb:7.1f}")
