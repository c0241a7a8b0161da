"""Seeded synthetic data for benchmarks and tests (no network, no datasets): checkpoints in the reference's state_dict layout,
images, and planted head outputs.  numpy PCG64 streams: identical in the build container and on the GPU box.
Pure data generation -- nothing here computes any part of the path."""
import math

import numpy as np
import torch

NAMES80 = [str(i) for i in range(80)]
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]           # reference models/YOLOPoint.py:21-28
STRIDES = (8.0, 16.0, 32.0)


def layout_of(model):
    return [(k, tuple(v.shape)) for k, v in model.state_dict().items()]


# ---------------------------------------------------------------------------------------------
# deterministic synthetic weights (numpy PCG64 streams: identical here and on the GPU box)
# ---------------------------------------------------------------------------------------------
# He-style gain of the synthetic conv weights per model size: 1.6 keeps the activations of the 25-layer-deep -n / -s O(1..40); the same
# gain compounds to 1e6 (-m) and 3e10 (-l) through their deeper C3 stacks -- beyond f16 -- so the deeper members get a smaller one
# (head logits O(1..500) at 640x640).  The golden vectors (-n / -s) use the default.
GAIN = {"n": 1.6, "s": 1.6, "m": 1.42, "l": 1.35, "x": 1.3}


def synth_state_dict(layout, seed, gain=1.6):
    """layout: list of (key, shape) in reference state_dict order -> fp32 tensors.  BN affine /
    running statistics are randomised so that BN folding is exercised."""
    sd = {}
    for idx, (key, shape) in enumerate(layout):
        rng = np.random.default_rng([seed, idx])
        shape = tuple(shape)
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.int64)
            continue
        if key.endswith("anchors"):
            a = torch.tensor(ANCHORS, dtype=torch.float32).view(3, 3, 2)
            sd[key] = a / torch.tensor(STRIDES).view(3, 1, 1)
            continue
        if key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean"):
            v = rng.normal(0.0, 0.1, shape)
        elif key.endswith("bn.weight"):
            v = rng.uniform(0.7, 1.3, shape)
        elif key.endswith("bn.bias") or key.endswith(".bias"):
            v = rng.normal(0.0, 0.1, shape)
        else:   # conv weight OIHW: He-style scale keeps activations O(1) through ~30 layers
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.normal(0.0, 1.0, shape) * (gain / math.sqrt(fan_in))
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return sd


def synth_image(B, C, H, W, seed):
    return torch.from_numpy(np.random.default_rng([seed, 777]).random((B, C, H, W), dtype=np.float32))


def make_model(version, seed, names=NAMES80, dtype="f32", model_name="YOLOPoint"):
    """Product model + the synthetic reference-layout state_dict loaded into it."""
    from .. import models
    m = models.Model(names=names, model_name=model_name, version=version)
    sd = synth_state_dict(layout_of(m), seed, gain=GAIN.get(version, 1.6))
    m.load_state_dict(sd, strict=True)
    m.set_compute_dtype(dtype)
    return m.eval(), sd


def planted_heatmap(H, W, npeaks, seed, noise=0.01, sigma=1.5):
    """Gaussian peaks of distinct heights over U(0, noise) background; every pixel value distinct."""
    rng = np.random.default_rng(seed)
    heat = rng.random((H, W)).astype(np.float64) * noise
    ys, xs = rng.integers(0, H, npeaks), rng.integers(0, W, npeaks)
    amp = rng.uniform(0.1, 0.9, npeaks)
    r = int(3 * sigma) + 1
    for y, x, a in zip(ys, xs, amp):
        y0, y1, x0, x1 = max(0, y - r), min(H, y + r + 1), max(0, x - r), min(W, x + r + 1)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        heat[y0:y1, x0:x1] += a * np.exp(-((yy - y) ** 2 + (xx - x) ** 2) / (2 * sigma ** 2))
    heat = (heat / heat.max() * 0.95).astype(np.float32)
    # make all fp32 values distinct so that sort order is total (parity is defined on distinct scores)
    flat = heat.ravel()
    order = np.argsort(flat, kind="stable")
    u = flat[order].copy()
    for i in range(1, len(u)):
        if u[i] <= u[i - 1]:
            u[i] = np.nextafter(u[i - 1], np.float32(2), dtype=np.float32)
    flat[order] = u
    return flat.reshape(H, W)


def planted_predictions(B, N, nc, ncand, seed, img=640):
    """[B,N,5+nc] decoded predictions: `ncand` rows per image with high objectness in overlapping clusters."""
    rng = np.random.default_rng(seed)
    p = np.zeros((B, N, 5 + nc), dtype=np.float32)
    p[..., 4] = rng.uniform(0.0, 0.2, (B, N))
    p[..., 5:] = rng.uniform(0.0, 0.3, (B, N, nc))
    p[..., 0:2] = rng.uniform(0, img, (B, N, 2))
    p[..., 2:4] = rng.uniform(8, 64, (B, N, 2))
    for b in range(B):
        rows = rng.choice(N, ncand, replace=False)
        nclust = max(1, ncand // 6)
        centres = rng.uniform(0.1 * img, 0.9 * img, (nclust, 2))
        sizes = np.exp(rng.uniform(np.log(0.03 * img), np.log(0.4 * img), (nclust, 2)))
        cl = rng.integers(0, nclust, ncand)
        p[b, rows, 0:2] = centres[cl] + rng.normal(0, 0.08, (ncand, 2)) * sizes[cl]
        p[b, rows, 2:4] = sizes[cl] * np.exp(rng.normal(0, 0.15, (ncand, 2)))
        p[b, rows, 4] = rng.uniform(0.3, 1.0, ncand)
        cls_main = rng.integers(0, nc, nclust)[cl]
        p[b, rows, 5 + cls_main] = rng.uniform(0.5, 1.0, ncand)
        second = (cls_main + 1 + rng.integers(0, max(nc - 1, 1), ncand)) % nc
        p[b, rows, 5 + second] = np.maximum(p[b, rows, 5 + second], rng.uniform(0.2, 0.9, ncand))
    return p


def planted_descriptors(D, N1, N2, frac, seed, noise=0.2):
    """Unit descriptors with `frac` planted correspondences (d2 = normalise(d1 + noise*g), permuted)."""
    rng = np.random.default_rng(seed)
    d1 = rng.normal(size=(D, N1)).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=0, keepdims=True)
    d2 = rng.normal(size=(D, N2)).astype(np.float32)
    nmatch = int(min(N1, N2) * frac)
    src = rng.choice(N1, nmatch, replace=False)
    dst = rng.choice(N2, nmatch, replace=False)
    d2[:, dst] = d1[:, src] + noise * rng.normal(size=(D, nmatch)).astype(np.float32) / np.sqrt(D)
    d2 /= np.linalg.norm(d2, axis=0, keepdims=True)
    return d1.astype(np.float32), d2.astype(np.float32)


def tracking_sequence(D, frames, N, seed, keep=0.7, noise=0.15):
    """A sequence of (pts [3,n] float64, desc [D,n] float32 unit columns): each frame carries `keep` of the previous frame's
    points over (descriptor + noise, position + 1 px, shuffled) and adds new ones; n varies per frame."""
    rng = np.random.default_rng(seed)
    out, prev = [], None
    for f in range(frames):
        n = int(N * rng.uniform(0.8, 1.2))
        d = rng.normal(size=(D, n)).astype(np.float32)
        xy = rng.uniform(8, 300, size=(2, n)).round()
        if prev is not None:
            pd, pxy = prev
            k = min(int(pd.shape[1] * keep), n)
            src = rng.choice(pd.shape[1], k, replace=False)
            dst = rng.choice(n, k, replace=False)
            d[:, dst] = pd[:, src] + noise * rng.normal(size=(D, k)).astype(np.float32) / np.sqrt(D)
            xy[:, dst] = pxy[:, src] + 1.0
        d /= np.linalg.norm(d, axis=0, keepdims=True)
        pts = np.vstack((xy, rng.uniform(0.02, 1.0, size=(1, n))))
        out.append((pts.astype(np.float64), d.astype(np.float32)))
        prev = (d, xy)
    return out
