"""Data parallelism for the training path: one process per GPU, replicated weights, per-rank BatchNorm statistics, bucketed
gradient all-reduce OVERLAPPED with the backward, gradient accumulation.

Reference: accelerate -> DistributedDataParallel(broadcast_buffers=False) (src/train.py:44-46,174,245): gradients are averaged
over ranks while `accelerator.backward(loss)` runs (DDP's reducer fires one all-reduce per bucket as the bucket's gradients
become final), BN buffers are never synchronised, `accelerator.accumulate` / `gas` (train.py:38-43,190) skips the all-reduce on
all but the last micro-batch of an optimizer step.  `torch.distributed` with backend "nccl" is RCCL over xGMI on MI355X; the
helper is backend-agnostic, so the same code runs under gloo in the CPU tests.

Schedule.  A training step back-propagates two forwards (train.py:208-245): the image pass through the whole network and the
warped pass through the keypoint / descriptor sub-graph only.  engine.TrainStep runs the FULL backward first: when it returns,
every parameter the second pass does not reach (Detect, PAN, YOLO encoder: ~87 % of YOLOPoint-s's gradient bytes) is final, and
`notify()` launches the all-reduce of those buckets -- asynchronously: ProcessGroupNCCL enqueues the collective on its own stream
behind an event recorded on the compute stream at that point -- while the keypoint-only backward keeps the compute stream busy.
The trunk / keypoint-head buckets follow when that pass returns; `finish()` makes the compute stream wait for the collectives
right before the optimizer reads the gradients.  The division by the world size is part of the collective (ReduceOp.AVG on RCCL;
gloo has no AVG: SUM and one scale per bucket).

Buckets hold parameters in gradient-ready order (training.grad_ready_groups), a few per group (default: 3 detector + 1 keypoint
bucket for YOLOPoint-s, 6-8 MB each: large enough to be bandwidth- rather than latency-bound on the 7-link xGMI mesh, small enough
to start early).  With `bind_grads()` (what TrainStep uses) the parameters' .grad ARE views of the buckets, so a step moves no
gradient bytes other than the collectives themselves; without it the gradients are copied in and out.
"""
import contextlib
from .switches import sw

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, params, group=None, bucket_bytes=None, groups=None, buckets_per_group=(3, 1)):
        """params: iterable of parameters (buckets in REVERSE registration order: the order a backward finishes them), or
        groups: [(name, [params in gradient-ready order]), ...] (training.grad_ready_groups): buckets never straddle a group."""
        self.group = group
        if groups is None:
            ps = [p for p in params if p.requires_grad]
            groups = [("all", list(reversed(ps)))]
            buckets_per_group = (max(1, sum(p.numel() for p in ps) * 4 // (bucket_bytes or (32 << 20)) + 1),) if bucket_bytes else (1,)
        self.params = [p for _, g in groups for p in g]
        self.buckets = []                       # list of (flat fp32 buffer, [(param, offset, numel)])
        self.bucket_group = []                  # group name per bucket
        self._views = None
        for gi, (gname, gparams) in enumerate(groups):
            total = sum(p.numel() for p in gparams)
            want = buckets_per_group[min(gi, len(buckets_per_group) - 1)]
            limit = bucket_bytes // 4 if bucket_bytes else -(-total // max(want, 1))
            cur, size = [], 0
            for p in gparams:
                n = p.numel()
                if cur and bucket_bytes and size + n > limit:      # byte budget: never exceed it
                    self._close(cur, size, gname)
                    cur, size = [], 0
                cur.append((p, size, n))
                size += -(-n // 4) * 4                             # (every slot starts 16-byte aligned: FlatAdam / vector loads)
                if not bucket_bytes and size >= limit:              # bucket count: close once the share is reached (no dangling remainder)
                    self._close(cur, size, gname)
                    cur, size = [], 0
            if cur:
                self._close(cur, size, gname)
        # all buckets are slices of ONE arena (bucket order), so that an optimizer can walk every gradient in one launch (optim.FlatAdam)
        dev = self.buckets[0][1][0][0].device
        sizes = [sz for sz, _ in self.buckets]
        self._arena_store = torch.zeros((sum(sizes) + 3) // 4 * 4, dtype=torch.float32, device=dev)     # (whole 16-byte units: yp_fill_zero)
        self.arena = self._arena_store[:sum(sizes)]
        self.bucket_offsets = [sum(sizes[:i]) for i in range(len(sizes))]
        self.buckets = [(self.arena[o:o + sz], entries) for o, (sz, entries) in zip(self.bucket_offsets, self.buckets)]
        self.param_arena = None
        self._bucket_of = {id(p): bi for bi, (_, entries) in enumerate(self.buckets) for p, _, _ in entries}
        self.expected = {id(p): 1 for p in self.params}     # gradient contributions per micro-batch (set_expected)
        self.sync = True
        self.force_collectives = False          # run the collectives even for a single rank (exercises the RCCL path on one GPU)
        import os
        comm = sw("YP_DP_COMM")
        if comm not in ("fp32", "bf16"):
            # (f16 is not offered: a SUM-reduced fp16 stage overflows at 65 504 on loss-scaled or large gradients; bf16 has fp32's range)
            raise ValueError(f"YP_DP_COMM={comm!r}: the gradient exchange runs in 'fp32' (default, as accelerate / DDP reduce in the reference) or 'bf16'")
        self.comm_dtype = torch.bfloat16 if comm == "bf16" else None      # None: fp32 buckets as they are
        self._stage = None
        self.launch_log = []                    # bucket indices in launch order (tests, bench reporting)
        self._pending, self._works, self._launched = None, {}, set()

    def _close(self, entries, size, gname):
        self.buckets.append((size, entries))                # (the flat buffers are cut from one arena at the end of __init__)
        self.bucket_group.append(gname)

    def flatten_parameters(self):
        """Move every parameter into one flat fp32 arena laid out exactly like the gradient arena (p.data becomes a view; values kept).
        Returns the arena.  The parameters' addresses change: plans built over the old ones are never used again (they are keyed by
        address) and the packed filters are re-derived."""
        if self.param_arena is None:
            arena = torch.zeros_like(self.arena)
            with torch.no_grad():
                for o, (_, entries) in zip(self.bucket_offsets, self.buckets):
                    for p, off, n in entries:
                        slot = arena[o + off:o + off + n].view_as(p)
                        slot.copy_(p.data)
                        p.data = slot
            self.param_arena = arena
            from .models.common import invalidate_packed_weights
            invalidate_packed_weights()
        return self.param_arena

    def grads_bound(self):
        return self._views is not None and all(self._bound(bi) for bi in range(len(self.buckets)))

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

    # -- zero-copy gradients ---------------------------------------------------------------------------------------------
    def bind_grads(self, zero=True):
        """Every p.grad becomes a view into its bucket (all fp32 parameters); with `zero` the buckets are cleared (one launch
        each).  Backward passes ACCUMULATE into them and the all-reduce runs in place -- no per-parameter copies.  Call instead of
        optimizer.zero_grad() at the start of an optimizer step (zero=False on the later micro-batches of an accumulation)."""
        if self._views is None:
            self._views = [[flat[off:off + n].view_as(p) for p, off, n in entries] for flat, entries in self.buckets]
        if zero:                                # (all buckets: one launch)
            if self.arena.is_cuda:
                from . import _hip
                _hip.check(_hip.lib().yp_fill_zero(self._arena_store.data_ptr(), self._arena_store.numel() * 4, _hip.stream_ptr()))
            else:
                self.arena.zero_()
        for (flat, entries), views in zip(self.buckets, self._views):
            for (p, _, _), v in zip(entries, views):
                if p.grad is not v:
                    p.grad = v

    def _bound(self, bi):
        views = self._views[bi] if self._views else None
        return views is not None and all(p.grad is v for (p, _, _), v in zip(self.buckets[bi][1], views))

    # -- overlapped reduction --------------------------------------------------------------------------------------------
    def set_expected(self, counts):
        """counts: {parameter: number of backward passes that contribute to it per micro-batch} (default 1 each)."""
        for p, c in counts.items():
            if id(p) in self.expected:
                self.expected[id(p)] = int(c)

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-batches of a gradient accumulation whose gradients are NOT reduced yet (DDP.no_sync / accelerator.accumulate)."""
        prev, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = prev

    def begin(self):
        """Start of a micro-batch's backward: every bucket waits for `expected` contributions to each of its parameters."""
        self._pending = [sum(self.expected[id(p)] for p, _, _ in entries) for _, entries in self.buckets]
        self._works, self._launched = {}, set()
        self.launch_log = []

    def notify(self, params):
        """A backward pass has accumulated its contribution to `params`.  Buckets whose parameters are all final are all-reduced
        NOW (asynchronously, behind the work queued on the current stream so far) unless inside no_sync()."""
        if self._pending is None:
            self.begin()
        hit = set()
        for p in params:
            bi = self._bucket_of.get(id(p))
            if bi is not None:
                self._pending[bi] -= 1
                hit.add(bi)
        for bi in sorted(hit):
            if self._pending[bi] <= 0:
                self._launch(bi)

    def _launch(self, bi):
        if bi in self._launched or not self.sync or (self.world == 1 and not self.force_collectives):
            return
        self._launched.add(bi)
        self.launch_log.append(bi)
        flat, entries = self.buckets[bi]
        if not self._bound(bi):                # copy mode: gather the gradients into the bucket first
            have = [(p, off, n) for p, off, n in entries if p.grad is not None]
            for p, off, n in entries:
                if p.grad is None:
                    flat[off:off + n].zero_()
            if have:
                torch._foreach_copy_([flat[off:off + n] for _, off, n in have], [p.grad.reshape(-1) for p, _, _ in have])
        avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        stage = None
        if self.comm_dtype is not None:
            # 16-bit exchange (YP_DP_COMM=bf16): the fp32 bucket is rounded into a staging buffer, the collective moves half the bytes over
            # xGMI, and finish() widens the rank average back into the fp32 bucket the optimizer reads (the accumulation over micro-batches
            # and Adam's arithmetic stay fp32; what is rounded is each rank's contribution to the average, once per optimizer step)
            if self._stage is None:
                self._stage = [torch.empty(f.numel(), dtype=self.comm_dtype, device=f.device) for f, _ in self.buckets]
            stage = self._stage[bi]
            if avg:
                stage.copy_(flat)
            else:                                  # SUM backends (gloo): the 1/world factor goes in BEFORE the rounding, the sum then stays a mean
                torch.mul(flat, 1.0 / self.world, out=flat)
                stage.copy_(flat)
        self._works[bi] = (dist.all_reduce(flat if stage is None else stage, op=op, group=self.group, async_op=True), avg, stage)

    def finish(self):
        """Launch whatever has not been launched (parameters without a gradient count as zeros), then make the current stream wait
        for every collective; the gradients are the rank average afterwards.  No-op inside no_sync() / for a single rank."""
        if not self.sync or (self.world == 1 and not self.force_collectives):
            self._pending = None
            return
        if self._pending is None:
            self.begin()
        for bi in range(len(self.buckets)):
            self._launch(bi)
        inv = 1.0 / self.world
        for bi, (work, avg, stage) in sorted(self._works.items()):
            work.wait()
            flat, entries = self.buckets[bi]
            if stage is not None:
                flat.copy_(stage)
            if not avg and stage is None:
                flat.mul_(inv)
            if self._bound(bi):
                continue
            for p, off, n in entries:
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone().to(p.dtype)
                else:
                    p.grad.copy_(g)
        self._pending = None

    def all_reduce(self):
        """Average p.grad over the ranks in one go (no overlap): begin + finish."""
        self.begin()
        self.finish()

    def payload_bytes(self):
        return sum(f.numel() * (4 if self.comm_dtype is None else 2) for f, _ in self.buckets)

    def describe(self):
        nb = 4 if self.comm_dtype is None else 2          # bytes per element on the wire
        return [{"group": g, "mbytes": round(f.numel() * nb / 1e6, 2), "params": len(e), "comm_dtype": "fp32" if nb == 4 else "bf16"}
                for g, (f, e) in zip(self.bucket_group, self.buckets)]


def shard_batch(n_global, rank, world):
    """Contiguous, even split of a global batch; the reference shards its loaders the same way (train.py:137)."""
    if n_global % world:
        raise ValueError(f"global batch {n_global} is not divisible by world size {world}")
    per = n_global // world
    return rank * per, (rank + 1) * per


def accumulation_steps(batch_per_device, n_devices, nominal=64):
    """gas of the reference (train.py:38-43): micro-batches per optimizer step so that the global batch is about `nominal`."""
    return max(round(nominal / (batch_per_device * n_devices)), 1)


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
