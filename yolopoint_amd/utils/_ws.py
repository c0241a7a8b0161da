"""Cached device scratch buffers for the post-processing entry points (caller-owned workspaces
of the C ABI; one growing uint8 tensor per (device, tag))."""
import torch

_cache = {}


def workspace(device, nbytes, tag="default"):
    key = (torch.device(device).index, tag)
    t = _cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _cache[key] = t
    return t


def as_cuda_f32(x, device=None, what="tensor"):
    """numpy / torch, any device -> contiguous fp32 cuda tensor (the H2D boundary of the path)."""
    import numpy as np
    from .. import _hip
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not isinstance(x, torch.Tensor):
        raise _hip.YpError(f"{what}: expected a numpy array or torch tensor, got {type(x).__name__}")
    if not x.is_cuda:
        _hip.require_gpu()
        x = x.to(device if device is not None else "cuda")
    return x.float().contiguous()
