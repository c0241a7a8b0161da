"""reference: src/models/model_wrap.py:410-606 (== demo.py:232-356) -- PointTracker: the mutual-NN descriptor matcher (HIP) and the
fixed-memory track bookkeeping around it (host, vectorised numpy: a few hundred ids per frame)."""
import numpy as np
import torch

from .. import _hip
from ..utils._ws import workspace, as_cuda_f32


class PointTracker(object):
    """tracks: M x (2 + max_length) rows [track id, running mean match distance, point id per remembered frame (-1 = none)];
    point ids index the concatenation of the remembered frames' point lists (get_offsets)."""

    def __init__(self, max_length=2, nn_thresh=0.7):
        if max_length < 2:
            raise ValueError('max_length must be greater than or equal to 2.')
        self.maxl = max_length
        self.nn_thresh = nn_thresh
        self.all_pts = [np.zeros((2, 0)) for _ in range(self.maxl)]
        self.last_desc = None
        self.tracks = np.zeros((0, self.maxl + 2))
        self.track_count = 0
        self.max_score = 9999
        self.matches = None
        self.last_pts = None
        self.mscores = None

    def get_offsets(self):
        """Start of each remembered frame's points in the global point numbering (reference :477-491)."""
        return np.cumsum(np.array([0] + [p.shape[1] for p in self.all_pts[:-1]]))

    def get_matches(self):
        return self.matches

    def get_mscores(self):
        return self.mscores

    def clear_desc(self):
        self.last_desc = None

    def update(self, pts, desc):
        """New frame: pts [3,N] (numpy or tensor), desc [D,N] (numpy or device tensor; kept where it is for the next match).
        Drops the oldest frame, matches against the previous one, extends matched tracks (running mean of the match distance)
        and opens a track per unmatched point (reference :503-577)."""
        if pts is None or desc is None:
            print('PointTracker: Warning, no points were added to tracker.')
            return
        pts = pts.detach().cpu().numpy() if isinstance(pts, torch.Tensor) else np.asarray(pts)
        assert pts.shape[1] == desc.shape[1]
        if self.last_desc is None:
            self.last_desc = np.zeros((desc.shape[0], 0))
        removed = self.all_pts[0].shape[1]
        self.all_pts = self.all_pts[1:] + [pts]
        t = np.delete(self.tracks, 2, axis=1)
        t[:, 2:] -= removed                               # ids of the remaining frames shift down; the dropped frame's go to -1
        t[:, 2:][t[:, 2:] < -1] = -1
        offsets = self.get_offsets()
        t = np.hstack((t, -np.ones((t.shape[0], 1))))
        matches = self.nn_match_two_way(self.last_desc, desc, self.nn_thresh)
        self.matches = matches
        if self.last_pts is not None:
            self.matches = np.concatenate((self.last_pts[:, matches[0].astype(int)], pts[:2, matches[1].astype(int)]), axis=0)
        matched = np.zeros(pts.shape[1], dtype=bool)
        if matches.shape[1] and t.shape[0]:
            id1 = matches[0].astype(np.int64) + offsets[-2]
            id2 = matches[1].astype(np.int64) + offsets[-1]
            prev = t[:, -2]
            order = np.argsort(prev, kind="stable")
            pos = np.clip(np.searchsorted(prev[order], id1), 0, len(order) - 1)
            rows = order[pos]
            hit = prev[rows] == id1                       # (a match is mutual: each previous point appears at most once)
            rows, id2h, dist = rows[hit], id2[hit], matches[2][hit]
            matched[matches[1].astype(np.int64)[hit]] = True
            t[rows, -1] = id2h
            fresh = t[rows, 1] == self.max_score
            length = (t[rows, 2:] != -1).sum(axis=1) - 1.0
            frac = 1.0 / np.maximum(length, 1.0)
            t[rows, 1] = np.where(fresh, dist, (1.0 - frac) * t[rows, 1] + frac * dist)
        new_ids = (np.arange(pts.shape[1]) + offsets[-1])[~matched]
        new = -np.ones((new_ids.shape[0], self.maxl + 2))
        new[:, -1] = new_ids
        new[:, 0] = self.track_count + np.arange(new_ids.shape[0])
        new[:, 1] = self.max_score
        t = np.vstack((t, new))
        self.track_count += new_ids.shape[0]
        self.tracks = t[np.any(t[:, 2:] >= 0, axis=1)]
        self.last_desc = desc.clone() if isinstance(desc, torch.Tensor) else desc.copy()
        self.last_pts = pts[:2, :].copy()

    def get_tracks(self, min_length):
        """Tracks observed in the newest frame with at least min_length observations (reference :579-596)."""
        if min_length < 1:
            raise ValueError("'min_length' too small.")
        keep = ((self.tracks[:, 2:] != -1).sum(axis=1) >= min_length) & (self.tracks[:, -1] != -1)
        return self.tracks[keep].copy()

    @_hip.guarded
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
