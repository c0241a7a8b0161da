"""CPU, world_size 2, gloo: the data-parallel plumbing (gradient bucketing + all-reduce, parameter broadcast,
batch sharding, the bench's barrier / max-over-ranks timing) is correct by construction."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolopoint_amd.dp import GradAllReducer, shard_batch, timed_region
    torch.manual_seed(100 + rank)                      # different initial weights per rank
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    red = GradAllReducer(net.parameters(), bucket_bytes=256)       # tiny buckets -> several collectives
    red.broadcast_parameters(net, src=0)
    w0 = net[0].weight.detach().clone()
    # rank-dependent gradients; one parameter has no gradient on rank 1
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    if rank == 1:
        net[2].bias.grad = None
    net[1].running_mean.fill_(float(rank))             # buffers must NOT be synchronised
    red.all_reduce()
    copy_grads = [p.grad.flatten().tolist() for p in net.parameters()]
    # zero-copy mode: .grad are views of the buckets, two "backward passes" accumulate into them, reduce in place
    red.bind_grads()
    views_ok = all(p.grad.data_ptr() != 0 and float(p.grad.abs().sum()) == 0.0 for p in net.parameters())
    for _ in range(2):
        for i, p in enumerate(net.parameters()):
            p.grad += float(rank + 1) * (i + 1)
    ptrs = [p.grad.data_ptr() for p in net.parameters()]
    red.all_reduce()
    bound = {"views_ok": views_ok, "same_storage": ptrs == [p.grad.data_ptr() for p in net.parameters()],
             "grads": [p.grad.flatten().tolist() for p in net.parameters()]}
    red.bind_grads()
    bound["rezeroed"] = all(float(p.grad.abs().sum()) == 0.0 for p in net.parameters())
    for i, p in enumerate(net.parameters()):           # restore the copy-mode results reported below
        p.grad = None
    lo, hi = shard_batch(64, rank, world)
    dt = timed_region(lambda: time.sleep(0.01 * (rank + 1)), steps=3, warmup=1, sync=lambda: None, barrier=dist.barrier,
                      reduce_max=lambda t: (lambda x: (dist.all_reduce(x, op=dist.ReduceOp.MAX), float(x))[1])(torch.tensor([t])))
    # plain Python payloads: torch tensors in an mp.Queue travel through shared-memory fds that die with the child
    out = {"w0": w0.flatten().tolist(), "grads": copy_grads, "bound": bound,
           "rm": net[1].running_mean.tolist(), "shard": (lo, hi), "nbuckets": len(red.buckets), "dt": dt}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["w0"] == b["w0"]                                                   # broadcast from rank 0
    assert a["nbuckets"] > 1
    n = len(a["grads"])
    for i, (ga, gb) in enumerate(zip(a["grads"], b["grads"])):
        assert ga == gb
        expect = (1 + 2) * (i + 1) / 2.0 if i != n - 1 else 1 * (i + 1) / 2.0  # last param: rank 1 contributed zeros
        assert all(abs(v - expect) < 1e-6 for v in ga), (i, ga[0], expect)
    for r in (a, b):
        assert r["bound"]["views_ok"] and r["bound"]["same_storage"] and r["bound"]["rezeroed"]
        for i, g in enumerate(r["bound"]["grads"]):                             # two accumulations of (rank+1)(i+1), averaged
            assert all(abs(v - 2 * (1 + 2) * (i + 1) / 2.0) < 1e-6 for v in g), (i, g[0])
    assert a["rm"][0] == 0.0 and b["rm"][0] == 1.0                             # per-rank BN statistics
    assert a["shard"] == (0, 32) and b["shard"] == (32, 64)
    assert abs(a["dt"] - b["dt"]) < 1e-9 and a["dt"] >= 0.06 - 1e-3            # max over ranks: rank 1's 3 x 20 ms
