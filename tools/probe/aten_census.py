#!/usr/bin/env python3
"""Which ATen operators does one steady-state training step still dispatch, and from where?  Runs a few steps, then one step under a
TorchDispatchMode that records every operator with the innermost yolopoint_amd / bench frame that caused it.  View-only operators are
listed separately (they launch nothing).  python tools/probe/aten_census.py [--version s --batch 16]"""
import argparse, collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from yolopoint_amd.utils.synthetic import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch

VIEWS = {"view", "_unsafe_view", "reshape", "expand", "permute", "transpose", "t", "slice", "select", "unsqueeze", "squeeze", "as_strided", "detach",
         "alias", "unbind", "split", "split_with_sizes", "chunk", "narrow", "flatten", "unflatten", "view_as", "_reshape_alias", "lift_fresh", "empty",
         "empty_like", "empty_strided", "new_empty", "_local_scalar_dense", "is_pinned", "record_stream", "size", "stride", "storage_offset", "numel",
         "is_same_size", "is_nonzero", "set_", "resize_", "item", "_to_copy_meta"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.where = collections.defaultdict(collections.Counter)
        self.syncs = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        site = "?"
        for fr in reversed(traceback.extract_stack()):
            fn = fr.filename
            if ("yolopoint_amd" in fn or fn.endswith("bench.py")) and "aten_census" not in fn:
                site = f"{os.path.relpath(fn, ROOT)}:{fr.lineno}"
                break
        self.ops[name] += 1
        self.where[name][site] += 1
        if name in ("_local_scalar_dense", "is_nonzero") and any(isinstance(x, torch.Tensor) and x.is_cuda for x in args):
            self.syncs[site] += 1           # a device value read back by the host
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--version", default="s"); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--size", type=int, default=640)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m, _ = make_model(a.version, 1, dtype="bf16")
    m = m.to(dev).train()
    step = TrainStep(m, dev, img_size=a.size)
    batch = synthetic_batch(a.batch, a.size, dev, 1234)
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize()
    c = Census()
    with c:
        step(batch)
    torch.cuda.synchronize()
    launching = {k: v for k, v in c.ops.items() if k not in VIEWS}
    print(f"# one training step: {sum(c.ops.values())} ATen dispatches, {sum(launching.values())} of them not view-only")
    print(f"# host read-backs of device values (synchronisations): {dict(c.syncs) if c.syncs else 'none'}"
          f"   (scalar reads of host tensors: {c.ops.get('_local_scalar_dense', 0) - sum(c.syncs.values())})")
    by_site = collections.Counter()
    for name, n in sorted(launching.items(), key=lambda kv: -kv[1]):
        sites = ", ".join(f"{s} x{k}" for s, k in c.where[name].most_common(6))
        print(f"{n:4d}  {name:28s} {sites}")
        for s, k in c.where[name].items():
            by_site[s] += k
    print("# by call site")
    for s, k in by_site.most_common(60):
        print(f"{k:4d}  {s}")


if __name__ == "__main__":
    main()
