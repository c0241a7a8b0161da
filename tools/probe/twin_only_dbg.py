"""Which parameters differ between YP_FP8_TWIN_ONLY=1 / 0 (and between two runs of the same mode) under a variant mixture.
YP_TUNE_ONLY=... YP_TUNE_RANDOM=n python tools/probe/twin_only_dbg.py [version] [pair]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import make_model
from yolopoint_amd.engine import TrainStep, synthetic_batch
from yolopoint_amd.models.common import invalidate_packed_weights
cuda = torch.device("cuda:0")
CFGS = [c.split(":") for c in (sys.argv[1] if len(sys.argv) > 1 else "s:0").split(",")]      # e.g. l:1,s:1,s:0 -- the test's order, one process
for version, pair in CFGS:
  os.environ["YP_TRAIN_PAIR"] = pair
  print("==", version, pair, flush=True)
  m0, _ = make_model(version, 23, dtype="bf16")
  m0 = m0.to(cuda).train()
  batches = [synthetic_batch(2, 128, cuda, 300 + i) for i in range(3)]
  res = []
  for mode in os.environ.get("MODES", "1,0,1,0").split(","):
      os.environ["YP_FP8_TWIN_ONLY"] = mode
      invalidate_packed_weights()
      if os.environ.get("POISON"):           # uninitialised reads pick this up: the caching allocator hands the freed blocks to the next graph
          g_ = [torch.full((64 << 20,), float(os.environ["POISON"]), device=cuda) for _ in range(8)]
          del g_
      m = copy.deepcopy(m0)
      step = TrainStep(m, cuda, img_size=128, lr=1e-3, fp8=os.environ.get("FP8", "1") == "1")
      step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
      losses = []
      for it in range(int(os.environ.get("STEPS", "3"))):
          torch.manual_seed(77 + it)
          losses.append(float(step(batches[it])))
      extra = {}
      for gs in m.model._train_graphs.values():
          for g in (gs if isinstance(gs, (list, tuple)) else [gs]):
              for hn, hd in getattr(g, "head_debug", {}).items():
                  for kk, vv in hd.items():
                      extra[f"dbg.{id(g) % 1000}.{hn}.{kk}"] = vv.buf.t.detach().float().clone()
      for n_, p_ in m.named_parameters():
          if p_.grad is not None:
              extra["grad." + n_] = p_.grad.detach().clone()
      res.append((mode, losses, {**{n: p.detach().clone() for n, p in m.named_parameters()}, **extra}))
      print(mode, losses, flush=True)
  for i in range(1, len(res)):
      a, b = res[0], res[i]
      bad = [(n, float((a[2][n] - b[2][n]).abs().max())) for n in a[2] if n in b[2] and a[2][n].shape == b[2][n].shape and not torch.equal(a[2][n], b[2][n])]
      print(f"run0(mode {a[0]}) vs run{i}(mode {b[0]}): {len(bad)} of {len(a[2])} parameters differ", bad[:6])
