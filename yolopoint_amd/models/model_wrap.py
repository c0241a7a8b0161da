"""reference: src/models/model_wrap.py:434-476 (== demo.py:300-341) — PointTracker's matcher."""
import numpy as np
import torch

from .. import _hip
from ..utils._ws import workspace, as_cuda_f32


class PointTracker(object):
    """Only the descriptor matcher of the reference tracker is on the hot path."""

    def __init__(self, max_length=2, nn_thresh=0.7):
        if max_length < 2:
            raise ValueError('max_length must be greater than or equal to 2.')
        self.maxl = max_length
        self.nn_thresh = nn_thresh
        self.mscores = None

    def nn_match_two_way(self, desc1, desc2, nn_thresh):
        """Mutual nearest-neighbour matching of unit descriptors desc1 [D,N1], desc2 [D,N2]
        -> float64 numpy [3, L] rows (idx1, idx2, L2 distance), idx1 ascending."""
        assert desc1.shape[0] == desc2.shape[0]
        if desc1.shape[1] == 0 or desc2.shape[1] == 0:
            return np.zeros((3, 0))
        assert nn_thresh > 0.0
        d1 = as_cuda_f32(desc1, what="desc1")
        d2 = as_cuda_f32(desc2, device=d1.device, what="desc2")
        D, N1 = d1.shape
        N2 = d2.shape[1]
        l = _hip.lib()
        out = torch.empty((3, N1), dtype=torch.float32, device=d1.device)
        cnt = torch.empty((1,), dtype=torch.int32, device=d1.device)
        ws = workspace(d1.device, l.yp_mnn_workspace_bytes(N1, N2), "mnn")
        _hip.check(l.yp_mnn_match(d1.data_ptr(), N1, d2.data_ptr(), N2, D, float(nn_thresh), out.data_ptr(),
                                  cnt.data_ptr(), N1, ws.data_ptr(), ws.numel(), _hip.stream_ptr()))
        L = int(cnt.item())
        matches = out[:, :L].cpu().numpy().astype(np.float64)
        self.mscores = matches
        return matches
