"""reference: src/utils/torch_utils_yolo.py:152-154 (de_parallel), :194-214 (fuse_conv_and_bn)."""
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
