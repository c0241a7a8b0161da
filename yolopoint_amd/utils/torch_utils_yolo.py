"""reference: src/utils/torch_utils_yolo.py:152-154 (de_parallel), :194-214 (fuse_conv_and_bn), :315-349 (ModelEMA)."""
import math
from copy import deepcopy

import torch
import torch.nn as nn


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def de_parallel(model):
    """Unwrap DP/DDP."""
    return model.module if is_parallel(model) else model


def fuse_conv_and_bn(conv, bn):
    """Return an nn.Conv2d (bias=True) whose weights absorb `bn`'s inference transform:
    W' = diag(gamma/sqrt(var+eps)) W,  b' = beta - gamma*mean/sqrt(var+eps) (+ scaled conv bias)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.eps + bn.running_var)
        fused.weight.copy_((torch.diag(scale) @ conv.weight.reshape(conv.out_channels, -1)).view(fused.weight.shape))
        b_conv = torch.zeros(conv.weight.size(0), device=conv.weight.device) if conv.bias is None else conv.bias
        b_bn = bn.bias - bn.weight * bn.running_mean / torch.sqrt(bn.running_var + bn.eps)
        fused.bias.copy_((torch.diag(scale) @ b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
    return fused


class ModelEMA:
    """Exponential moving average of everything in a model's state_dict (reference :315-349: same constructor, `.ema`, `.updates`,
    `update(model)`, `update_attr(model)`): ema <- d * ema + (1 - d) * model with the ramped decay d = decay * (1 - exp(-updates / 2000)).
    The reference loops over the ~290 tensors with two small kernels each; here every floating-point tensor of the two state_dicts goes
    through ONE multi-tensor lerp (ema + (1 - d) * (model - ema)), and the plans packed from the EMA weights are told they are stale."""

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(de_parallel(model)).eval()          # fp32 EMA (native plans / graphs are per-object caches and are not copied)
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            src = de_parallel(model).state_dict()
            mine, theirs = [], []
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    mine.append(v)
                    theirs.append(src[k].detach().to(v.dtype))
            torch._foreach_lerp_(mine, theirs, 1.0 - d)
        from ..models.common import invalidate_packed_weights
        invalidate_packed_weights()

    def update_attr(self, model, include=(), exclude=('process_group', 'reducer')):
        """Copy plain attributes (names, hyp, ...) from the model to the EMA copy (reference copy_attr, :303-309)."""
        for k, v in model.__dict__.items():
            if (len(include) and k not in include) or k.startswith('_') or k in exclude:
                continue
            setattr(self.ema, k, v)
