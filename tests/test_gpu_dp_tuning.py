"""Data-parallel replicas share ONE kernel-variant table: rank 0 times the candidates of a convolution signature and broadcasts its choice
(yolopoint_amd/plan.py::_autotune), so that every rank of a job runs the same kernels (same fp32 summation orders, same step time).
Two processes on this one GPU (gloo carries the two-float broadcast; NCCL refuses two ranks on one device) build the same small plan."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolopoint_amd import _hip, plan
        from yolopoint_amd.plan import PlanBuilder
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        choices = []
        for cin, cout, k, H in ((128, 128, 3, 40), (256, 256, 1, 20), (64, 128, 1, 40)):
            pb = PlanBuilder(4, _hip.YP_F16, dev)
            x = pb.new_buf(H, H, cin)
            x.t.normal_()
            g = torch.Generator().manual_seed(cin + k)
            pb.conv(x.view(), torch.randn(cout, cin, k, k, generator=g) * 0.05, torch.zeros(cout), k, 1, k // 2, _hip.YP_ACT_SILU)
            p = pb.finish()
            p.run()
            torch.cuda.synchronize()
        choices = sorted((str(k_), v[0]) for k_, v in plan._TUNE_CACHE.items())
        q.put((rank, choices))
    finally:
        dist.destroy_process_group()


def test_two_ranks_run_the_same_kernel_variants(cuda):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1] and len(got[0]) == 3, got


def _step_worker(rank, world, port, q, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)     # RCCL: one device per rank; gloo: both ranks on this one GPU
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from helpers import make_model
        from yolopoint_amd.engine import TrainStep, synthetic_batch
        m, _ = make_model("n", 5, dtype="bf16")                 # same initial weights on every rank
        m = m.to(dev).train()
        step = TrainStep(m, dev, img_size=128)
        step.sparse = dict(num_samples_per_image=100, num_masked_non_matches_per_match=30)
        losses = []
        for it in range(2):
            torch.manual_seed(100 + it)                          # same sampling draws on every rank; the data differ
            losses.append(float(step(synthetic_batch(2, 128, dev, 1000 * rank + it))))
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().float().flatten() for p in m.parameters()])
        q.put((rank, losses, float(flat.double().sum()), float(flat.double().abs().sum()), list(step.reducer.launch_log),
               getattr(step.reducer, "exposed_ms", None)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_train_in_lockstep(cuda):
    """engine.TrainStep (native loss stage, overlapped bucketed reducer) with two ranks on this one GPU over gloo: different data per rank,
    averaged gradients -> bit-identical parameters on both ranks after two optimizer steps; the buckets were launched detector group first."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (l0, s0, a0, log0, _), (l1, s1, a1, log1, _) = got[0], got[1]
    assert l0 != l1                                   # different batches
    assert s0 == s1 and a0 == a1, (s0, s1, a0, a1)    # the same weights after the same averaged updates
    assert log0 == log1 and len(log0) > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL refuses two ranks on one device: needs >= 2 visible GPUs")
def test_two_ranks_train_in_lockstep_over_rccl():
    """The same two-rank step over RCCL (backend "nccl"), one MI355X per rank: what `bench.py --gpus N` runs.  Parameters bit-identical on
    both ranks, the bucket launch order non-empty and equal, and the reducer reports its exposed wait (HIP events around finish())."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=900) for _ in procs)}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (l0, s0, a0, log0, ex0), (l1, s1, a1, log1, ex1) = got[0], got[1]
    assert l0 != l1 and s0 == s1 and a0 == a1 and log0 == log1 and len(log0) > 0
    assert ex0 is None or isinstance(ex0, (int, float))


def test_bench_rehearsal_two_ranks_share_this_gpu():
    """`bench.py --gpus 2 --rehearsal`: the WHOLE N > 1 control flow of bench.py -- self-launch under torch.distributed.run, process group, replica
    timing with barrier + max over ranks, rank 0's shared autotuning broadcasts, the data-parallel `train` record with the real bucketed gradient
    all-reduce and its exposed-communication events, rank 0 alone printing ONE line -- on real kernels, two ranks sharing this GPU over gloo
    (RCCL refuses two ranks on one device; with >= 2 GPUs the driver's `--gpus N` runs the same code over RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearsal", "--only", "train", "--steps", "6", "--warmup", "2",
                        "--train-steps", "3", "--train-warmup", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "replicas" and "rehearsal" in rec
    tr = rec["train"]
    assert tr["n_gpus"] == 2 and tr["parallelism"] == "dp2" and tr["global_batch"] == 16
    assert tr["bucket_launch_order"] == [0, 1, 2, 3], tr["bucket_launch_order"]          # detector buckets behind the YOLO-branch plan, keypoint bucket last
    assert isinstance(tr["exposed_comm_ms_per_step"], float)
    assert "cpu_baseline" not in rec and "parity" not in tr                                   # (rank 0, N = 1 only)
