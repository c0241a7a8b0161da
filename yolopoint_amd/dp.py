"""Data parallelism for the training path: one process per GPU, replicated weights, per-rank BatchNorm
statistics, ONE bucketed gradient all-reduce per optimizer step.

Reference: accelerate -> DistributedDataParallel(broadcast_buffers=False) (src/train.py:44-46,174,245):
gradients are averaged over ranks, BN buffers are never synchronised, rank 0's buffers are what a
checkpoint holds.  `torch.distributed` with backend "nccl" is RCCL over xGMI on MI355X; the helper is
backend-agnostic, so the same code runs under gloo in the CPU tests.

Bucketing: parameters are packed in REVERSE registration order (descriptor/keypoint heads and Detect
first, Conv1 last = the order in which the backward plan finishes their gradients) into few large flat
fp32 buffers (default 32 MB: YOLOPoint-s is one bucket of 30.6 MB, -l seven) so that each collective is
bandwidth- rather than latency-bound on the 7-link xGMI mesh.  `all_reduce()` launches every bucket
asynchronously, then waits and scales.  With `bind_grads()` (what TrainStep uses) the parameters' .grad ARE views of the
buckets, so a step moves no gradient bytes other than the collective itself; without it the gradients are copied in and out.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, params, group=None, bucket_bytes=32 << 20):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                       # list of (flat fp32 buffer, [(param, offset, numel)])
        self._views = None
        cur, size = [], 0
        for p in reversed(self.params):
            n = p.numel()
            if cur and (size + n) * 4 > bucket_bytes:
                self._close(cur, size)
                cur, size = [], 0
            cur.append((p, size, n))
            size += n
        if cur:
            self._close(cur, size)

    def _close(self, entries, size):
        dev = entries[0][0].device
        self.buckets.append((torch.zeros(size, dtype=torch.float32, device=dev), entries))

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def broadcast_parameters(self, module, src=0):
        """Rank `src`'s parameters (not buffers: broadcast_buffers=False) become everyone's, as DDP does at construction."""
        if self.world == 1:
            return
        for p in module.parameters():
            dist.broadcast(p.data, src=src, group=self.group)
        # a write through .data bumps neither Tensor._version nor the optimizer-step count: tell the plans their packed filters are stale
        from .models.common import invalidate_packed_weights
        invalidate_packed_weights()

    def bind_grads(self):
        """Zero-copy mode: every p.grad becomes a view into its bucket (all fp32 parameters), the buckets are cleared with one
        launch each, backward passes ACCUMULATE into them and all_reduce() reduces them in place -- no per-parameter copies.
        Call instead of optimizer.zero_grad() at the start of a step."""
        if self._views is None:
            self._views = [[flat[off:off + n].view_as(p) for p, off, n in entries] for flat, entries in self.buckets]
        for (flat, entries), views in zip(self.buckets, self._views):
            flat.zero_()
            for (p, _, _), v in zip(entries, views):
                if p.grad is not v:
                    p.grad = v

    def _bound(self, entries, views):
        return views is not None and all(p.grad is v for (p, _, _), v in zip(entries, views))

    def all_reduce(self):
        """Average p.grad over the ranks (missing gradients count as zeros)."""
        if self.world == 1:
            return
        works, bound = [], []
        for i, (flat, entries) in enumerate(self.buckets):
            is_bound = self._bound(entries, self._views[i] if self._views else None)
            bound.append(is_bound)
            if not is_bound:
                have = [(p, off, n) for p, off, n in entries if p.grad is not None]
                for p, off, n in entries:
                    if p.grad is None:
                        flat[off:off + n].zero_()
                if have:
                    torch._foreach_copy_([flat[off:off + n] for _, off, n in have], [p.grad.reshape(-1) for p, _, _ in have])
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        inv = 1.0 / self.world
        for (flat, entries), w, is_bound in zip(self.buckets, works, bound):
            w.wait()
            flat.mul_(inv)
            if is_bound:
                continue
            for p, off, n in entries:
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone().to(p.dtype)
                else:
                    p.grad.copy_(g)

    def payload_bytes(self):
        return sum(f.numel() * 4 for f, _ in self.buckets)


def shard_batch(n_global, rank, world):
    """Contiguous, even split of a global batch; the reference shards its loaders the same way (train.py:137)."""
    if n_global % world:
        raise ValueError(f"global batch {n_global} is not divisible by world size {world}")
    per = n_global // world
    return rank * per, (rank + 1) * per


def timed_region(step, steps, warmup, sync, barrier, reduce_max):
    """The measurement contract of bench.py: W untimed steps, then exactly K steps bracketed by barrier + device sync
    on both sides; returns the MAX wall time over ranks."""
    import time
    for _ in range(warmup):
        step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(); barrier(); sync()
    return reduce_max(time.perf_counter() - t0)
