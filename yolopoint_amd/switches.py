"""Every `YP_*` environment switch of the package, in ONE registry.

`sw(name)` is the only way the package reads a switch: an unregistered name raises (tests/test_host_layout.py checks that no other
`YP_*` environment read exists in the sources), and at import time any `YP_*` variable in the environment that is not registered raises
as well -- a mistyped switch can no longer silently select the default.

Kinds:
  path   selects an alternative code path that must compute the same thing.  `alt` lists the non-default values and `scope`
         (`infer` / `train` / `fp8`) the harness that runs it: tests/test_gpu_switches.py runs EVERY registered (name, alt value) through a small
         forward or optimizer step and compares with the default run at `tol` (0 = bit-identical; otherwise relative L2 of every head /
         parameter gradient).  A path switch without a test cannot exist: the harness iterates this table.
  knob   a number that changes scheduling / grid sizes only (results do not depend on it; the knobs with a results-independence test name it).
  debug  diagnostics and test hooks (printing, explicit kernel-variant selection, library override).
  native read by libyolopoint_hip.so itself (getenv in csrc/), listed here so that the unknown-variable check knows them.

Switches that selected "measured and dropped" formulations without a test were removed in round 6 (YP_ADAM_FUSED, YP_AUX_STREAM, YP_WGRAD_DET,
YP_WGRAD_GROUP, YP_PACK_BATCH, YP_BN_EPILOGUE, YP_STEM_WGRAD, YP_FP8_FUSE, YP_BN_BWD_RES, YP_MERGE_SIBLINGS, YP_SAMPLE_SORTED, YP_FUSE_ONLY_C,
YP_TUNE_RANDOM_LIMIT, YP_PLAN_DEBUG, YP_TRAIN_BWD_LANES (its gradients differed from the default's), YP_WG_BLOCK64 / _CAP / _TARGET / _NOSTORE, YP_STREAM_PICK): their defaults are now the only behaviour.
"""
import os
from collections import namedtuple

Switch = namedtuple("Switch", "default kind doc alt scope tol")


def _s(default, kind, doc, alt=(), scope=None, tol=0.0):
    return Switch(default, kind, doc, tuple(alt), scope, tol)


SWITCHES = {
    # ---- inference plan: fusions and lanes (yolopoint_amd/models/*.py, plan.py)
    "YP_FUSE_STEM": _s("1", "path", "fused stem kernel reading the NCHW fp32 image (0: pack_input + generic Conv1)", ["0"], "infer", 2e-3),
    "YP_FUSE_STEM2": _s("1", "path", "Conv1 + Conv2 in one launch (0: two launches)", ["0"], "infer", 2e-3),
    "YP_FUSE_STEM3": _s("1", "path", "... + Bottleneck1.cv1/cv2 in the same launch (0: separate pointwise launch)", ["0"], "infer", 2e-3),
    "YP_FUSE_BOTTLENECK": _s("1", "path", "fused Bottleneck kernel (0: cv1 and cv2 as two launches; auto: where the tuner times it faster)", ["0", "auto"], "infer", 2e-3),
    "YP_FUSE_C3_TAIL": _s("1", "path", "C3's cv3 inside the last fused Bottleneck launch (0: its own launch)", ["0"], "infer", 2e-3),
    "YP_INFER_LANES": _s("1", "path", "heads / Detect levels on the plan's side lane (0: one lane)", ["0"], "infer", 0.0),
    "YP_LANES_EAGER": _s("1", "path", "two-lane plans replay eagerly on two streams (0: captured into a hipGraph with a side branch)", ["0"], "infer", 0.0),
    "YP_HEADS_FORK": _s("4", "path", "the heads are forked behind Bottleneck<2|3|4>", ["2", "3"], "infer", 0.0),
    "YP_GRAPH_LINEAR": _s("0", "path", "1: captured graphs keep the linear capture chain (no dependency rewiring)", ["1"], "infer", 0.0),
    # ---- training graph (training.py)
    "YP_TRAIN_PAIR": _s("1", "path", "both forwards of a step as one 2B-sample pass (0: two graphs, autograd loss stage; bf16 noise between the two: 0.09 median)", ["0"], "train", 0.15),
    "YP_TRAIN_LANES": _s("0", "path", "1: weight-gradient kernels on a second lane, per-layer launches with fp32 atomics (2e-6 absolute)", ["1"], "train", 1e-4),
    "YP_TRAIN_FWD_LANES": _s("1", "path", "forward plan heads on the side lane (0: one lane)", ["0"], "train", 0.0),
    "YP_TRAIN_DET_LANES": _s("1", "path", "Detect levels 0 / 1 of the training forward on the side lane", ["0"], "train", 0.0),
    "YP_TRAIN_PARALLEL": _s("0", "path", "1: training plans replay as dependency DAGs instead of linear chains", ["1"], "train", 0.0),
    "YP_TRAIN_GRAPH": _s("1", "path", "replay the training launch lists as hipGraphs (0: eager; fwd / bwd: one side only)", ["0", "fwd", "bwd"], "train", 0.0),
    "YP_DGRAD_PHASES": _s("1", "path", "stride-2 dgrad as four parity-class convolutions (0: zero-stuffed 3x3)", ["0"], "train", 2e-2),
    # ---- loss stage (engine.py, utils/loss_functions.py)
    "YP_NATIVE_STAGE": _s("1", "path", "native loss stage between the plans (0: torch autograd over the loss API)", ["0"], "train", 1e-3),
    "YP_NATIVE_PREPARE": _s("1", "path", "InfoNCE sampling by device kernels (0: the PyTorch formulation; other random stream)", ["0"], "train_draws", 0.0),
    "YP_NATIVE_INFONCE": _s("1", "path", "InfoNCE kernels (0: torch formulation inside the autograd stage; fp32 rows where the kernels of a bf16 graph gather bf16 rows)", ["0"], "train_autograd", 1e-3),
    "YP_NATIVE_DETLOSS": _s("1", "path", "detector-loss kernel (0: torch formulation inside the autograd stage)", ["0"], "train_autograd", 1e-4),
    "YP_PREPARE_SYNC": _s("0", "path", "1: InfoNCE sampling with one host read-back of the pool size", ["1"], "train", 0.0),
    "YP_TRAIN_SIDE_STREAM": _s("1", "path", "label work / loss lanes on the side stream (0: everything on the main stream)", ["0"], "train", 0.0),
    "YP_LOSS_LANES": _s("2", "path", "2: InfoNCE chain beside the YOLO-branch backward; 1: object + detector losses beside InfoNCE; 0: one stream", ["1", "0"], "train", 0.0),
    "YP_LABELS_ORDER": _s("split", "path", "enqueue order of the label kernels on the side lane (split | first | after)", ["first", "after"], "train", 0.0),
    "YP_NCE_ROWS": _s("bf16", "path", "InfoNCE gathers over bf16 rows in bf16 graphs (fp32: fp32 rows)", ["fp32"], "train", 2e-2),
    "YP_ADAM": _s("flat", "path", "optim.FlatAdam, one launch (torch: torch.optim.Adam(fused=True))", ["torch"], "train_adam", 2e-2),
    "YP_DP_COMM": _s("fp32", "path", "gradient exchange dtype (bf16: buckets travel as bf16); tests/test_dp_gloo.py", [], None),
    # ---- fp8 mode (training.py)
    "YP_FP8_FWD": _s("1", "path", "fp8 forward convolutions (0: bf16 forward inside an fp8 graph)", ["0"], "fp8", 0.5),
    "YP_FP8_DGRAD": _s("1", "path", "fp8 dgrad (0: bf16 dgrad)", ["0"], "fp8", 0.5),
    "YP_FP8_WGRAD": _s("1", "path", "8-bit weight gradients (0: 16-bit)", ["0"], "fp8", 0.5),
    "YP_FP8_TWIN_ONLY": _s("1", "path", "unread 16-bit BatchNorm copies are not written (0: every copy kept)", ["0"], "fp8", 0.0),
    # ---- knobs (results do not depend on them)
    "YP_SIDE_WGS": _s("256", "knob", "grid cap of the label kernels on the side lane (test_capped_grids_give_the_same_results)"),
    "YP_NCE_WGS": _s("", "knob", "grid cap of the InfoNCE gathers beside a backward plan (default by D / dtype)"),
    "YP_STEPS_IN_FLIGHT": _s("2", "knob", "optimizer steps the host may queue ahead (test_loss_stage_lanes_and_the_bound_on_queued_steps)"),
    "YP_TUNE_ITERS": _s("8", "knob", "timed launches per autotuner candidate"),
    "YP_TUNE_COLD": _s("0", "knob", "MB swept through the L2s in front of every timed autotuner launch"),
    "YP_TUNE_SHARED": _s("1", "knob", "rank 0 tunes and broadcasts (0: per-rank tuning); tests/test_gpu_dp_tuning.py"),
    "YP_FP8_MARGIN": _s("1.0", "knob", "headroom factor of the delayed fp8 scales"),
    "YP_FP8_AMAX_SLOTS": _s("", "knob", "sub-slots of the amax atomics"),
    # ---- debug / test hooks
    "YP_HIP_LIB": _s("", "debug", "path of another build of libyolopoint_hip.so (A/B runs, probe builds)"),
    "YP_TUNE_ONLY": _s("", "debug", "restrict the autotuner to these variant ids"),
    "YP_TUNE_RANDOM": _s("", "debug", "stress mode: a pseudo-random applicable variant per signature (seed)"),
    "YP_TUNE_FORCE": _s("", "debug", "with YP_TUNE_RANDOM: explicit signature-index:variant pairs"),
    "YP_TUNE_DEBUG": _s("", "debug", "print the autotuner's timings / picks"),
    "YP_BENCH_RCCL_N1": _s("1", "debug", "bench.py: the one-rank RCCL leg of the N = 1 train record (0: skip)"),
    "YP_PROFILE_LOSS": _s("", "debug", "tools/: loss-stage profiling hook"),
    "YP_TEST_SPLITK": _s("", "debug", "tests/test_gpu_model.py: force split-K tiles on the plan-unique layers"),
    # ---- read by the native library (csrc/)
    "YP_NCE_GEN": _s("2", "native", "InfoNCE gather generation (1: first generation; tests/test_gpu_training.py runs both)"),
    "YP_STREAM_DEBUG": _s("", "native", "print yp_stream_pick's concurrency tests"),
    "YP_GRAPH_DUMP": _s("", "native", "print the captured graph topology"),
    "YP_MMA8_PROBE": _s("", "native", "probe build (make probe8) elimination switches"),
    "YP_WSK_PROBE": _s("", "native", "probe build (make probewsk) elimination switches"),
}


def sw(name):
    """Value of a registered switch (string; the registered default when unset)."""
    try:
        d = SWITCHES[name]
    except KeyError:
        raise KeyError(f"{name} is not a registered switch (yolopoint_amd/switches.py)") from None
    return os.environ.get(name, d.default)


def on(name):
    """True unless the switch is "0" (the convention of the on-by-default path switches)."""
    return sw(name) != "0"


def check_environment(environ=None):
    """Raise on `YP_*` variables that are not registered (a mistyped switch must not silently select the default)."""
    env = os.environ if environ is None else environ
    unknown = sorted(k for k in env if k.startswith("YP_") and k not in SWITCHES)
    if unknown:
        raise RuntimeError(f"unknown YP_* environment variable(s) {unknown}: see yolopoint_amd/switches.py for the registered switches")


check_environment()
