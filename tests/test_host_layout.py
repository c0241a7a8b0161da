"""CPU: host-side mirror of the reference interface — checkpoint key layout, parameter order,
fuse(), load_state_dict tolerance, weight packing, and the C-ABI surface (no compute calls)."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from helpers import NAMES80, make_model
from yolopoint_amd import _hip, models
from yolopoint_amd.plan import pack_conv_weight
from yolopoint_amd.utils import utils as U
from yolopoint_amd.utils.general_yolo import make_divisible, xywh2xyxy
from yolopoint_amd.utils.torch_utils_yolo import fuse_conv_and_bn, de_parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lay():
    with open(os.path.join(G, "layouts.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("v", ["n", "s", "m", "l"])
def test_state_dict_layout_matches_reference(lay, v):
    m = models.Model(names=NAMES80, model_name="YOLOPoint", version=v)
    mine = [[k, list(t.shape)] for k, t in m.state_dict().items()]
    assert mine == lay[v]["state_dict"]
    assert sum(p.numel() for p in m.parameters()) == lay[v]["n_params"]
    assert [k for k, _ in m.named_parameters()] == lay[v]["named_parameters"]     # freeze_layers() indices


@pytest.mark.parametrize("v", ["n", "s"])
def test_v52_state_dict_layout_matches_reference(lay, v):
    m = models.Model(names=NAMES80, model_name="YOLOPointv52", version=v)
    assert [[k, list(t.shape)] for k, t in m.state_dict().items()] == lay["v52_" + v]["state_dict"]
    assert sum(p.numel() for p in m.parameters()) == lay["v52_" + v]["n_params"]
    assert [k for k, _ in m.named_parameters()] == lay["v52_" + v]["named_parameters"]


def test_fused_and_single_class_layouts(lay):
    m = models.Model(names=NAMES80, version="n").eval().fuse()
    assert [[k, list(t.shape)] for k, t in m.state_dict().items()] == lay["n_fused"]["state_dict"]
    m1 = models.Model(names=["car"], version="n")
    assert [[k, list(t.shape)] for k, t in m1.state_dict().items()] == lay["n_nc1"]["state_dict"]
    assert models.Model(names=(), version="n").model.Detect.nc == 1           # empty names -> 1 dummy class


def test_detect_attributes_and_bias_init():
    m = models.Model(names=NAMES80, version="s")
    d = m.model.Detect
    assert (d.nc, d.no, d.nl, d.na) == (80, 85, 3, 3)
    assert torch.equal(d.stride, torch.tensor([8., 16., 32.]))
    assert torch.allclose(d.anchors[0, 0], torch.tensor([10 / 8, 13 / 8]))
    assert torch.allclose(d.anchors[2, 2], torch.tensor([373 / 32, 326 / 32]))
    fresh = models.YOLOPoint(0.5, 0.33, 3, 80, models.YOLOPoint.__init__.__defaults__ and [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]])
    torch.manual_seed(0)
    # obj bias shifted by log(8/(640/s)^2), cls by log(0.6/(nc-0.999999))  (reference YOLOPoint.py:92-100)
    b = d.m[0].bias.view(3, -1)
    assert abs(float(b[:, 5:].mean()) - np.log(0.6 / (80 - 0.999999))) < 0.2
    assert abs(float(b[:, 4].mean()) - np.log(8 / (640 / 8) ** 2)) < 0.2


def test_load_state_dict_tolerance():
    src = models.Model(names=NAMES80, version="n")
    dst = models.Model(names=["a", "b"], version="n")                  # different class count
    before = dst.state_dict()["model.Conv2.conv.weight"].clone()
    dst.load_state_dict(src.state_dict(), strict=True)                 # must not raise: partial load
    after = dst.state_dict()
    assert torch.equal(after["model.Conv2.conv.weight"], src.state_dict()["model.Conv2.conv.weight"])
    assert not torch.equal(after["model.Conv2.conv.weight"], before)
    assert after["model.Detect.m.0.bias"].shape == (21,)
    with pytest.raises(Exception):
        models.Model(names=NAMES80, version="q")


def test_freeze_layers_and_versions():
    m = models.Model(names=NAMES80, version="n")
    m.freeze_layers([0, 1, 2], verbose=False)
    ps = list(m.parameters())
    assert not ps[0].requires_grad and not ps[2].requires_grad and ps[3].requires_grad
    assert make_divisible(33, 8) == 40 and make_divisible(64 * 0.75, 8) == 48
    assert de_parallel(m) is m
    assert U.load_model(names=NAMES80, model_name="YOLOPoint", version="n").model.Conv1.conv.weight.shape == (16, 3, 6, 6)


def test_fuse_conv_and_bn_algebra():
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(8, 12, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(12, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1); bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    x = torch.randn(2, 8, 7, 7)
    f = fuse_conv_and_bn(conv, bn)
    assert torch.allclose(f(x), bn(conv(x)), atol=1e-5)


def test_pack_conv_weight_layout():
    """[Cout][Kpad] rows, k = (r*S + s)*Cin + c, zero padded (CPU check of the layout the kernel reads)."""
    w = torch.arange(5 * 8 * 3 * 3, dtype=torch.float32).view(5, 8, 3, 3)
    b = torch.arange(5, dtype=torch.float32)
    wp, bp, Kpad, Npad = pack_conv_weight(w, b, _hip.YP_F32, "cpu")
    assert (Kpad, Npad) == (96, 8) and wp.shape == (9, 96)      # + the zero row behind the filter
    assert float(wp[3, (1 * 3 + 2) * 8 + 5]) == float(w[3, 5, 1, 2])
    assert float(wp[:, 72:].abs().max()) == 0 and float(wp[5:].abs().max()) == 0 and torch.equal(bp[:5], b)
    wp16, _, Kpad16, _ = pack_conv_weight(w, None, _hip.YP_F16, "cpu")
    assert Kpad16 == 128 and wp16.dtype == torch.float16


def test_box_format_helper():
    x = np.array([[10., 20., 4., 6.]], dtype=np.float32)
    assert np.array_equal(xywh2xyxy(x), np.array([[8., 17., 12., 23.]], dtype=np.float32))
    assert torch.equal(xywh2xyxy(torch.from_numpy(x)), torch.tensor([[8., 17., 12., 23.]]))


def test_labels2Dto3D_roundtrip():
    post = np.load(os.path.join(G, "postproc.npz"))
    out = U.labels2Dto3D(torch.from_numpy(post["l2d.labels"]))
    np.testing.assert_allclose(out.numpy(), post["l2d.out"])
    np.testing.assert_allclose(U.getMasks(torch.from_numpy(post["l2d.mask"]), "cpu").numpy(), post["l2d.maskout"])


# ------------------------------------------------------------------------------ C ABI surface
def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "yolopoint_hip.h")).read()
    declared = set(re.findall(r"\b(yp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    l = _hip.lib()                                   # dlopen works without a GPU
    for name in declared:
        assert hasattr(l, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (yp_[a-z0-9_]+)", nm))
    assert declared <= exported
    assert l.yp_version() >= 100 and l.yp_conv_kpad(27, _hip.YP_F16) == 64 and l.yp_conv_kpad(144, _hip.YP_F32) == 160


def test_cabi_struct_sizes_and_argument_errors():
    assert ctypes.sizeof(_hip.YpView) == 32 and ctypes.sizeof(_hip.YpConvDesc) == 5 * 32 + 16 + 23 * 4 + 4 + 16 + 16 + 16 + 8 + 8 + 16 + 16 + 8 + 24 + 16       # (+ the fused-stem fields)
    l = _hip.lib()
    # argument validation happens before any device work, so it is testable on a CPU-only host
    rc = l.yp_conv2d(None, None)
    assert rc == -1 and b"null descriptor" in l.yp_last_error()
    d = _hip.YpConvDesc()
    d.dtype = 7
    assert l.yp_conv2d(ctypes.byref(d), None) == -1 and b"bad dtype" in l.yp_last_error()
    assert l.yp_kp_nms_workspace_bytes(1, 640, 640) > 640 * 640 * 9
    assert l.yp_mnn_workspace_bytes(1000, 2000) >= 8 * 3000


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under yolopoint_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "yolopoint_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)


def test_cpu_inputs_fail_loudly():
    m, _ = make_model("n", 1)
    with pytest.raises(_hip.YpError):
        m(torch.zeros(1, 3, 64, 64))
    if _hip.lib().yp_device_count() == 0:
        with pytest.raises(_hip.YpError):
            U.getPtsFromHeatmap(np.zeros((16, 16), np.float32), 0.1, 4)


def test_graph_schedule_respects_dependencies():
    """Pure host logic: graph edges = RAW/WAR/WAW conflicts on overlapping buffer slices, nothing else."""
    from yolopoint_amd.plan import PlanBuilder
    pb = PlanBuilder.__new__(PlanBuilder)
    pb.accesses = []
    # an access = (allocation, buffer base, lo, hi, first byte, last byte): same buffer -> channel ranges, slices of one arena -> byte ranges
    A, B_, Cc, D = 0x1000, 0x2000, 0x3000, 0x4000
    acc = lambda base, lo, hi, alloc=None, nbytes=0x800: (alloc or base, base, lo, hi, base, base + nbytes)
    pb.accesses.append(([], [acc(A, 0, 64)]))                           # 0: produce x
    pb.accesses.append(([acc(A, 0, 64)], [acc(B_, 0, 32)]))             # 1: cv1(x)  -> cat[0:32]
    pb.accesses.append(([acc(A, 0, 64)], [acc(B_, 32, 64)]))            # 2: cv2(x)  -> cat[32:64]  (independent of 1)
    pb.accesses.append(([acc(B_, 0, 64)], [acc(Cc, 0, 64)]))            # 3: cv3(cat) needs both
    pb.accesses.append(([acc(Cc, 0, 64)], [acc(Cc, 0, 64)]))            # 4: in-place on the result
    pb.accesses.append(([acc(A, 0, 64)], [acc(D, 0, 8)]))               # 5: another reader of x
    assert pb.dependencies() == [[], [0], [0], [1, 2], [3], [0]]
    # slices of one arena (weight-gradient accumulators) against a whole-arena clear: byte ranges decide; redundant edges are dropped
    pb.accesses = []
    ARENA = 0x9000
    pb.accesses.append(([], [acc(ARENA, 0, 1 << 30, ARENA, 0x1000)]))                   # 0: clear the arena
    pb.accesses.append(([], [acc(ARENA + 0x100, 0, 8, ARENA, 0x100)]))                  # 1: writes slice [0x100, 0x200)
    pb.accesses.append(([], [acc(ARENA + 0x200, 0, 8, ARENA, 0x100)]))                  # 2: writes slice [0x200, 0x300): independent of 1
    pb.accesses.append(([acc(ARENA + 0x100, 0, 8, ARENA, 0x100), acc(ARENA + 0x200, 0, 8, ARENA, 0x100)], [acc(D, 0, 8)]))   # 3: reads both
    assert pb.dependencies() == [[], [0], [0], [1, 2]]                 # (3 -> 0 is implied by 1 and 2)


def test_model_ema_matches_reference_formula():
    """utils/torch_utils_yolo.py:315-349: ema <- d ema + (1 - d) model, d = decay (1 - exp(-updates / 2000)), every floating-point state_dict
    tensor (parameters and BN statistics), integer buffers untouched; deepcopy of a model leaves its native plan caches behind."""
    import math
    from copy import deepcopy
    from yolopoint_amd.utils.torch_utils_yolo import ModelEMA
    m, _ = make_model("n", 4)
    m.model.__dict__["_plans"] = {"sentinel": object()}
    ema = ModelEMA(m, decay=0.99)
    assert "_plans" not in ema.ema.model.__dict__ and not any(p.requires_grad for p in ema.ema.parameters())
    before = {k: v.clone() for k, v in ema.ema.state_dict().items()}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.1)
        for b in m.buffers():
            if b.dtype.is_floating_point:
                b.add_(0.05)
    for upd in (1, 2):
        ema.update(m)
        d = 0.99 * (1 - math.exp(-upd / 2000))
        now = m.state_dict()
        for k, v in ema.ema.state_dict().items():
            if v.dtype.is_floating_point:
                want = before[k] * d + (1 - d) * now[k]
                assert torch.allclose(v, want, rtol=1e-5, atol=1e-7), k
                before[k] = want
            else:
                assert torch.equal(v, before[k])
    assert ema.updates == 2
    deepcopy(m)


def test_flat_parameter_arena_keeps_values_and_puts_c3_siblings_back_to_back():
    """dp.GradAllReducer.flatten_parameters() + training.link_siblings (what engine.TrainStep does to a model): every parameter becomes a
    view of one arena laid out like the gradient arena (values, state_dict keys and shapes unchanged, 16-byte aligned slots), and cv1 /
    cv2 of every C3 block -- filters, BatchNorm weights, biases and running statistics -- lie back to back, which is what lets the training
    plans run the two layers as one (TrainGraph.merged_siblings).  CPU tensors: pure host logic."""
    from yolopoint_amd.dp import GradAllReducer
    from yolopoint_amd.training import grad_ready_groups, link_siblings
    m, sd = make_model("n", 9)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    red = GradAllReducer(None, groups=grad_ready_groups(m.model))
    arena = red.flatten_parameters()
    link_siblings(m.model)
    assert arena.numel() == red.arena.numel() and arena.numel() % 4 == 0
    after = m.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    base = arena.untyped_storage().data_ptr()
    for p in m.parameters():
        assert p.data.untyped_storage().data_ptr() == base and (p.data_ptr() - arena.data_ptr()) % 16 == 0
    red.bind_grads()
    for (flat, entries), off in zip(red.buckets, red.bucket_offsets):      # the gradient views mirror the parameter slots
        for p, o, n in entries:
            assert p.grad.data_ptr() - red.arena.data_ptr() == p.data_ptr() - arena.data_ptr() == 4 * (off + o)
    n_c3 = 0
    for mod in m.model.modules():
        if type(mod).__name__ != "C3":
            continue
        n_c3 += 1
        for a, b in ((mod.cv1.conv.weight, mod.cv2.conv.weight), (mod.cv1.bn.weight, mod.cv2.bn.weight), (mod.cv1.bn.bias, mod.cv2.bn.bias),
                     (mod.cv1.bn.running_mean, mod.cv2.bn.running_mean), (mod.cv1.bn.running_var, mod.cv2.bn.running_var)):
            assert a.data_ptr() + 4 * a.numel() == b.data_ptr()
    assert n_c3 == 10
    # the bucket plan still separates the groups: detector buckets first, then the keypoint group
    assert red.bucket_group == sorted(red.bucket_group, key=lambda g: g != "detector")


def test_plan_key_sees_a_same_shape_module_replacement():
    """The cached inference plan is keyed on the parameters' identities: assigning a NEW module of the same shape (net.ConvDescA = ...)
    must change the key -- a key built on child COUNTS kept replaying the plan packed from the old module's filters."""
    from yolopoint_amd.models import common
    from helpers import make_model
    m, _ = make_model("n", 3)
    net = m.model
    v0 = net._weights_version()
    assert net._weights_version() == v0
    net.ConvDescA = common.Conv(net.ConvDescA.conv.in_channels, net.ConvDescA.conv.out_channels, 3, 2)
    v1 = net._weights_version()
    assert v1 != v0
    net.Detect.m[1] = torch.nn.Conv2d(net.Detect.m[1].in_channels, net.Detect.m[1].out_channels, 1)
    assert net._weights_version() != v1


def test_bbox_iou_public_helper_matches_the_oracle_ciou():
    from yolopoint_amd.utils.metrics_yolo import bbox_iou
    from oracle import loss_oracle
    g = torch.Generator().manual_seed(2)
    a, b = torch.rand(64, 4, generator=g) + 0.05, torch.rand(64, 4, generator=g) + 0.05
    assert float((bbox_iou(a, b, CIoU=True).squeeze(-1) - loss_oracle.ciou(a, b)).abs().max()) < 1e-6
    iou = bbox_iou(a, a)
    assert float((iou - 1).abs().max()) < 1e-4          # (eps = 1e-7 in the union of boxes with area ~1e-2)


def test_every_environment_switch_is_registered():
    """yolopoint_amd/switches.py is the only place the package reads `YP_*` variables: no `os.environ` / `getenv` access to a YP_ name elsewhere
    in the Python sources, every getenv("YP_...") of the native sources is registered, sw() refuses unknown names, and an unregistered YP_*
    variable in the environment is an error at import."""
    import glob
    import re
    from yolopoint_amd import switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "yolopoint_amd", "**", "*.py"), recursive=True):
        if path.endswith("switches.py"):
            continue
        src = open(path).read()
        assert not re.search(r"environ[^\n]*YP_|getenv\([^\n]*YP_", src), f"{path}: reads a YP_* variable outside switches.sw()"
        for name in re.findall(r"\bsw\(\"(YP_[A-Z0-9_]+)\"\)|\bon\(\"(YP_[A-Z0-9_]+)\"\)", src):
            assert (name[0] or name[1]) in switches.SWITCHES, (path, name)
    for path in glob.glob(os.path.join(root, "yolopoint_amd", "csrc", "*.h*")):
        for name in re.findall(r"getenv\(\"(YP_[A-Z0-9_]+)\"\)", open(path).read()):
            assert name in switches.SWITCHES and switches.SWITCHES[name].kind == "native", (path, name)
    for name in re.findall(r"YP_[A-Z0-9_]+", open(os.path.join(root, "bench.py")).read()):
        if name.startswith(("YP_BENCH", "YP_PROFILE")):
            assert name in switches.SWITCHES, name
    with pytest.raises(KeyError):
        switches.sw("YP_NO_SUCH_SWITCH")
    with pytest.raises(RuntimeError):
        switches.check_environment({"YP_TRIAN_PAIR": "0"})
    for name, s_ in switches.SWITCHES.items():
        assert s_.kind in ("path", "knob", "debug", "native"), name
        if s_.kind == "path" and name != "YP_DP_COMM":
            assert s_.alt and s_.scope, f"{name}: a path switch needs its alternative values and the harness scope that tests them"


def test_bitonic_network_of_the_csr_bucket_sort_sorts():
    """csrc/sampling.hip::wave_bitonic256 restated in numpy (64 lanes x 4 registers, element e = 4 * lane + r; strides 1 / 2 inside a lane,
    strides >= 4 across lanes with partner lane ^ (stride / 4); direction bit = e & size): every stage schedule of the 36 stages sorts ascending
    over e, with INT_MAX padding behind k <= 256 entries -- the rule the device code implements (GPU: tests/test_gpu_sampling.py)."""
    rng = np.random.default_rng(7)
    lane = np.arange(64)

    def cx(x, y, asc):
        mn, mx = np.minimum(x, y), np.maximum(x, y)
        return np.where(asc, mn, mx), np.where(asc, mx, mn)
    for k in (1, 2, 3, 63, 64, 65, 200, 255, 256):
        for _ in range(5):
            vals = rng.permutation(1 << 20)[:k].astype(np.int64)
            flat = np.full(256, 0x7fffffff, dtype=np.int64)
            flat[:k] = vals
            v = flat.reshape(64, 4).copy()
            size = 2
            while size <= 256:
                stride = size >> 1
                while stride > 0:
                    if stride >= 4:
                        ls = stride >> 2
                        keep_min = ((lane & ls) == 0) == (((4 * lane) & size) == 0)
                        o = v[lane ^ ls]
                        v = np.where(keep_min[:, None], np.minimum(v, o), np.maximum(v, o))
                    elif stride == 2:
                        asc = ((4 * lane) & size) == 0
                        v[:, 0], v[:, 2] = cx(v[:, 0].copy(), v[:, 2].copy(), asc)
                        v[:, 1], v[:, 3] = cx(v[:, 1].copy(), v[:, 3].copy(), asc)
                    elif size == 2:
                        v[:, 0], v[:, 1] = cx(v[:, 0].copy(), v[:, 1].copy(), np.ones(64, bool))
                        v[:, 2], v[:, 3] = cx(v[:, 2].copy(), v[:, 3].copy(), np.zeros(64, bool))
                    else:
                        asc = ((4 * lane) & size) == 0
                        v[:, 0], v[:, 1] = cx(v[:, 0].copy(), v[:, 1].copy(), asc)
                        v[:, 2], v[:, 3] = cx(v[:, 2].copy(), v[:, 3].copy(), asc)
                    stride >>= 1
                size <<= 1
            out = v.reshape(-1)
            assert np.array_equal(out[:k], np.sort(vals)) and np.all(out[k:] == 0x7fffffff), k
