"""CPU fp32 restatement of the YOLOPoint network forward.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py): never imported by the product path.

Written as pure functions over a reference-layout state_dict (fp32 OIHW tensors, keys
`model.<Block>...`), using torch's CPU conv / batch_norm as the arithmetic substrate — the same
substrate the reference itself runs on (ATen CPU), since this is a floating-point path whose
parity bar is a tolerance, not bit equality.  Pinned against golden outputs of the imported
reference: tests/golden/make_golden.py, tests/test_oracle_golden.py.

Each function cites the reference code it restates (paths relative to /root/reference/src).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-3, 0.03          # models/common.py:18-20

VERSIONS = {'n': (0.33, 0.25), 's': (0.33, 0.5), 'm': (0.67, 0.75), 'l': (1., 1.), 'x': (1.33, 1.25)}   # models/YOLOPoint.py:36-49
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]           # models/YOLOPoint.py:11-15
STRIDES = (8.0, 16.0, 32.0)


def arch(version):
    """Channel widths and C3 repeat counts (models/YOLOPoint.py:152-153)."""
    dm, wm = VERSIONS[version]
    c = [int(math.ceil(2 ** k * wm / 8) * 8) for k in range(6, 11)]
    n = [max(round(k * dm), 1) for k in (3, 6, 9)]
    return c, n


# Test hook: when set to a callable (name, x, w) -> (x, w), every Conv block passes its input and filter through it before the
# convolution -- the fp8 parity test installs per-tensor e4m3 fake quantisation here (BASELINE.json configs[4] has no reference
# implementation: the oracle for it is this restatement + the quantisation rule stated in csrc/fp8.hip).
FAKE_QUANT = None


def conv_block(sd, p, x, k, s, pad, training=False, stats=None):
    """Conv: SiLU(BN(conv(x))) or, after fuse(), SiLU(conv(x)+b)   (models/common.py:22-34)."""
    w = sd[p + ".conv.weight"]
    if FAKE_QUANT is not None:
        x, w = FAKE_QUANT(p, x, w)
    if p + ".bn.weight" in sd:
        y = F.conv2d(x, w, None, s, pad)
        rm, rv = sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"]
        if training:      # batch statistics; running stats updated in place on clones kept in `stats`
            rm, rv = rm.clone(), rv.clone()
            y = F.batch_norm(y, rm, rv, sd[p + ".bn.weight"], sd[p + ".bn.bias"], True, BN_MOMENTUM, BN_EPS)
            if stats is not None:
                stats[p + ".bn.running_mean"], stats[p + ".bn.running_var"] = rm, rv
        else:
            y = F.batch_norm(y, rm, rv, sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, BN_MOMENTUM, BN_EPS)
    else:
        y = F.conv2d(x, w, sd[p + ".conv.bias"], s, pad)
    return F.silu(y)


def bottleneck(sd, p, x, **kw):
    """x + cv2(cv1(x)), cv1 1x1, cv2 3x3 (models/common.py:79-89; always with the add inside C3)."""
    return x + conv_block(sd, p + ".cv2", conv_block(sd, p + ".cv1", x, 1, 1, 0, **kw), 3, 1, 1, **kw)


def c3(sd, p, x, n, **kw):
    """cv3(cat(m(cv1(x)), cv2(x)))  (models/common.py:123-135)."""
    a = conv_block(sd, p + ".cv1", x, 1, 1, 0, **kw)
    for i in range(n):
        a = bottleneck(sd, f"{p}.m.{i}", a, **kw)
    b = conv_block(sd, p + ".cv2", x, 1, 1, 0, **kw)
    return conv_block(sd, p + ".cv3", torch.cat((a, b), 1), 1, 1, 0, **kw)


def sppf(sd, p, x, **kw):
    """cv2(cat(x, m(x), m(m(x)), m(m(m(x))))), m = MaxPool2d(5,1,2)  (models/common.py:213-229)."""
    x = conv_block(sd, p + ".cv1", x, 1, 1, 0, **kw)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    return conv_block(sd, p + ".cv2", torch.cat((x, y1, y2, y3), 1), 1, 1, 0, **kw)


def detect(sd, p, feats, nc, training):
    """Detect head (models/yolo.py:49-81): 1x1 conv -> [B,na,ny,nx,no]; eval adds the sigmoid /
    grid / anchor decode and concatenates the levels to [B, sum(na*ny*nx), no]."""
    no, na = nc + 5, 3
    anchors = sd[p + ".anchors"]            # [nl, na, 2] in grid units
    xs, z = [], []
    for i, f in enumerate(feats):
        y = F.conv2d(f, sd[f"{p}.m.{i}.weight"], sd[f"{p}.m.{i}.bias"])
        B, _, ny, nx = y.shape
        y = y.view(B, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        xs.append(y)
        if not training:
            yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing='ij')
            grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2).float()
            ag = (anchors[i] * STRIDES[i]).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2).float()
            s = y.sigmoid()
            xy = (s[..., 0:2] * 2 - 0.5 + grid) * STRIDES[i]
            wh = (s[..., 2:4] * 2) ** 2 * ag
            z.append(torch.cat((xy, wh, s[..., 4:]), -1).view(B, -1, no))
    return xs if training else (torch.cat(z, 1), xs)


def yolopoint_forward(sd, x, version, nc=80, training=False, stats=None):
    """models/YOLOPoint.py:198-246.  Returns {'semi','desc','objects'} like the reference."""
    (c1, c2, c3_, c4, c5), (n1, n2, n3) = arch(version)
    kw = dict(training=training, stats=stats)
    P = "model."
    x = conv_block(sd, P + "Conv1", x, 6, 2, 2, **kw)
    x = conv_block(sd, P + "Conv2", x, 3, 2, 1, **kw)
    xa = c3(sd, P + "Bottleneck1", x, n1, **kw)
    x8 = conv_block(sd, P + "Conv3", xa, 3, 2, 1, **kw)
    semi = F.conv2d(c3(sd, P + "BottleneckDet", x8, n1, **kw), sd[P + "ConvDet.weight"])
    xb = c3(sd, P + "Bottleneck2", x8, n2, **kw)
    dA = conv_block(sd, P + "ConvDescA", xa, 3, 2, 1, **kw)
    dB = F.interpolate(conv_block(sd, P + "ConvDescB", xb, 3, 2, 1, **kw), scale_factor=2, mode='nearest')
    desc = F.conv2d(c3(sd, P + "BottleneckDesc", torch.cat((dA, dB), 1), n1, **kw), sd[P + "ConvDesc.weight"], None, 1, 1)
    desc = desc / torch.norm(desc, p=2, dim=1).unsqueeze(1)          # no epsilon (YOLOPoint.py:219-220)
    x = conv_block(sd, P + "Conv4", xb, 3, 2, 1, **kw)
    xc = c3(sd, P + "Bottleneck3", x, n3, **kw)
    x = conv_block(sd, P + "Conv5", xc, 3, 2, 1, **kw)
    x = c3(sd, P + "Bottleneck4", x, n1, **kw)
    x = sppf(sd, P + "SPPooling", x, **kw)
    xd = conv_block(sd, P + "Conv6", x, 1, 1, 0, **kw)
    x = c3(sd, P + "Bottleneck5", torch.cat((F.interpolate(xd, scale_factor=2, mode='nearest'), xc), 1), n1, **kw)
    xe = conv_block(sd, P + "Conv7", x, 1, 1, 0, **kw)
    xf = c3(sd, P + "Bottleneck6", torch.cat((F.interpolate(xe, scale_factor=2, mode='nearest'), xb), 1), n1, **kw)
    x = conv_block(sd, P + "Conv8", xf, 3, 2, 1, **kw)
    xg = c3(sd, P + "Bottleneck7", torch.cat((x, xe), 1), n1, **kw)
    x = conv_block(sd, P + "Conv9", xg, 3, 2, 1, **kw)
    p5 = c3(sd, P + "Bottleneck8", torch.cat((x, xd), 1), n1, **kw)
    return {'semi': semi, 'desc': desc, 'objects': detect(sd, P + "Detect", [xf, xg, p5], nc, training)}


def c2f(sd, p, x, n, **kw):
    """C2f: cv2(cat(chunk(cv1(x), 2) + [m_i(prev)])), m_i = two 3x3 Convs without shortcut (models/common.py:91-103,151-171)."""
    y = list(conv_block(sd, p + ".cv1", x, 1, 1, 0, **kw).chunk(2, 1))
    for i in range(n):
        t = conv_block(sd, f"{p}.m.{i}.cv1", y[-1], 3, 1, 1, **kw)
        y.append(conv_block(sd, f"{p}.m.{i}.cv2", t, 3, 1, 1, **kw))
    return conv_block(sd, p + ".cv2", torch.cat(y, 1), 1, 1, 0, **kw)


def yolopointv52_forward(sd, x, version, nc=80, training=False, stats=None):
    """models/YOLOPoint.py:294-342 (YOLOPointv52)."""
    (c1, c2, c3_, c4, c5), (n1, n2, n3) = arch(version)
    kw = dict(training=training, stats=stats)
    P = "model."
    up = lambda t: F.interpolate(t, scale_factor=2, mode='nearest')
    x = conv_block(sd, P + "Conv1", x, 6, 2, 2, **kw)
    x = conv_block(sd, P + "Conv2", x, 3, 2, 1, **kw)
    xa = c2f(sd, P + "Bottleneck1", x, n1, **kw)
    x8 = conv_block(sd, P + "Conv3", xa, 3, 2, 1, **kw)
    semi = c2f(sd, P + "BottleneckDet", x8, n1, **kw)
    xb = c2f(sd, P + "Bottleneck2", x8, n2, **kw)
    dA = F.max_pool2d(xa, 2, 2)
    dB = up(conv_block(sd, P + "ConvDescB", xb, 3, 2, 1, **kw))
    desc = c2f(sd, P + "BottleneckDesc", torch.cat((dA, dB), 1), n1, **kw)
    desc = desc / torch.norm(desc, p=2, dim=1).unsqueeze(1)
    x = conv_block(sd, P + "Conv4", xb, 3, 2, 1, **kw)
    xc = c2f(sd, P + "Bottleneck3", x, n3, **kw)
    x = conv_block(sd, P + "Conv5", xc, 3, 2, 1, **kw)
    x = c2f(sd, P + "Bottleneck4", x, n1, **kw)
    xd = sppf(sd, P + "SPPooling", x, **kw)
    xe = c2f(sd, P + "Bottleneck5", torch.cat((up(xd), xc), 1), n1, **kw)
    xf = c2f(sd, P + "Bottleneck6", torch.cat((up(xe), xb), 1), n1, **kw)
    x = conv_block(sd, P + "Conv8", xf, 3, 2, 1, **kw)
    xg = c2f(sd, P + "Bottleneck7", torch.cat((x, xe), 1), n1, **kw)
    x = conv_block(sd, P + "Conv9", xg, 3, 2, 1, **kw)
    p5 = c2f(sd, P + "Bottleneck8", torch.cat((x, xd), 1), n1, **kw)
    return {'semi': semi, 'desc': desc, 'objects': detect(sd, P + "Detect", [xf, xg, p5], nc, training)}


def fuse_conv_bn(w, gamma, beta, mean, var, eps=BN_EPS):
    """utils/torch_utils_yolo.py:194-214: W' = diag(g/sqrt(v+eps)) W ; b' = beta - g*mean/sqrt(v+eps)."""
    scale = gamma / torch.sqrt(eps + var)
    return w * scale.view(-1, 1, 1, 1), beta - gamma * mean / torch.sqrt(var + eps)


# ---------------------------------------------------------------------------------------------
# the 16-bit noise floor of the network itself (what PyTorch's own half-precision inference differs from fp32 by)
# ---------------------------------------------------------------------------------------------
class half_storage:
    """Context manager: inside it every convolution of this module rounds its input and its filter to `dtype` (fp16 | bf16) and
    accumulates in fp32 -- the arithmetic of PyTorch's half-precision inference (16-bit storage of activations and BN-folded filters,
    fp32 accumulation), stated on the CPU.  The difference between this forward and the plain fp32 oracle is the FLOOR any 16-bit
    implementation of the network sits on; the HIP path's 16-bit outputs are held to a small multiple of it
    (tests/test_gpu_bench_shapes.py) instead of to free-standing bars."""

    def __init__(self, dtype=torch.float16):
        self.dtype = dtype

    def __enter__(self):
        import torch.nn.functional as F0
        dt = self.dtype

        class Shim:
            def __getattr__(self, name):
                return getattr(F0, name)

            @staticmethod
            def conv2d(x, w, b=None, *a, **k):
                return F0.conv2d(x.to(dt).float(), w.to(dt).float(), b, *a, **k)
        global F
        self.saved = F
        F = Shim()
        return self

    def __exit__(self, *exc):
        global F
        F = self.saved
        return False


def fused_state_dict(sd):
    """BN folded into every Conv block of a reference-layout state_dict (utils/torch_utils_yolo.py:194-214 through the oracle's
    fuse_conv_bn): `.conv.weight` / `.conv.bias`, no `.bn.*` keys -- the tensors the inference path rounds to 16 bits."""
    out = {}
    for k, v in sd.items():
        if ".bn." in k:
            continue
        if k.endswith(".conv.weight") and k[:-len(".conv.weight")] + ".bn.weight" in sd:
            p = k[:-len(".conv.weight")]
            w, b = fuse_conv_bn(v, sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"])
            out[k], out[p + ".conv.bias"] = w, b
        else:
            out[k] = v
    return out


# ---------------------------------------------------------------------------------------------
# deterministic synthetic weights / images: data generators shared with the benchmarks (yolopoint_amd/utils/synthetic.py)
# ---------------------------------------------------------------------------------------------
from yolopoint_amd.utils.synthetic import synth_state_dict, synth_image  # noqa: E402,F401


# ---------------------------------------------------------------------------------------------
# backward golden vectors: seeded output projections as the loss, gradient sketches
# ---------------------------------------------------------------------------------------------
def output_projections(o, seed):
    """Seeded N(0,1) projections with the shapes of the three train-mode outputs ({'semi','desc','objects': list[3]})."""
    g = torch.Generator().manual_seed(1000 + seed)
    return {"semi": torch.randn(o["semi"].shape, generator=g), "desc": torch.randn(o["desc"].shape, generator=g),
            "objects": [torch.randn(t.shape, generator=g) for t in o["objects"]]}


def projected_loss(o, proj, device="cpu"):
    """loss = 0.01 <semi, P_semi> + <desc, P_desc> + 0.01 sum_i <objects_i, P_i>  (the scalar whose backward is the golden vector)."""
    l = (o["semi"] * proj["semi"].to(device)).sum() * 0.01 + (o["desc"] * proj["desc"].to(device)).sum()
    for t, p in zip(o["objects"], proj["objects"]):
        l = l + (t * p.to(device)).sum() * 0.01
    return l


def sign_projections(name, size, n):
    """n seeded +-1 vectors of length `size` (float64 [n, size]); the stream is keyed on the tensor's name so that the generator of
    the golden file and the tests agree without storing the vectors."""
    import zlib
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) + 77))
    return rng.integers(0, 2, (n, size), dtype=np.int8).astype(np.float64) * 2.0 - 1.0
