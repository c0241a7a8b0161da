"""reference: src/evaluations/descriptor_evaluation.py:148-181 (== demo.py:200-215)."""
import numpy as np
import torch

from .. import _hip
from ..utils._ws import as_cuda_f32


@_hip.guarded
def sample_desc_from_points(coarse_desc, pts, device=None, cell_size=8):
    """Bilinear-sample the coarse descriptor map [D,Hc,Wc] (or [1,D,Hc,Wc]) at pts [3,N] / [2,N]
    (x, y in full-resolution pixels) and L2-normalise each column -> float32 numpy [D, N]."""
    if coarse_desc.dim() != 4:
        coarse_desc = coarse_desc.view(*(1,) * (4 - coarse_desc.dim()), *coarse_desc.shape)
    D, Hc, Wc = coarse_desc.shape[1], coarse_desc.shape[2], coarse_desc.shape[3]
    pts = np.asarray(pts)
    if pts.ndim != 2 or pts.shape[1] == 0:
        return np.empty((D, 0))
    d = coarse_desc if coarse_desc.is_cuda and coarse_desc.dtype == torch.float32 else as_cuda_f32(coarse_desc.detach(), device)
    N = pts.shape[1]
    xy = torch.from_numpy(np.ascontiguousarray(pts[:2, :].T.astype(np.float32))).to(d.device)
    out = torch.empty((D, N), dtype=torch.float32, device=d.device)
    _, sc, sy, sx = d.stride()
    _hip.check(_hip.lib().yp_desc_sample(d.data_ptr(), D, Hc, Wc, sc, sy, sx, xy.data_ptr(), N, int(cell_size),
                                         out.data_ptr(), _hip.stream_ptr()))
    return out.cpu().numpy()
