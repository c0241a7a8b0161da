import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "statistical: the bar is a statistic of a noisy quantity (training curves, trained checkpoints); collected last")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch.device("cuda:0")


@pytest.fixture
def fixed_kernel_variants(monkeypatch):
    """fp32 gradient-parity tests against the oracle: every convolution runs its default kernel variant instead of the one the plan-time
    autotuner happens to time fastest.  Max pooling (SPPF, 4 x 4 maps in these small cases) routes a gradient to the FIRST maximum of its
    window; two activations that differ by less than the fp32 noise of a different summation order (another tile of an upstream layer) can
    swap places, and every gradient upstream of the pool then differs from the oracle's by a percent.  Measured: one such near-tie in the
    data of test_pair_pass_matches_two_oracle_passes[...-True] flipped with 2 of 25 variant mixtures (tools/probe/pair_grad_errors.py,
    YP_TUNE_FORCE=3:5,5:3; other data seeds: none) -- a property of the comparison, not of a kernel: every variant passes on its own
    (YP_TUNE_ONLY) and the variants themselves are covered by tests/test_gpu_blocks.py / test_gpu_conv_mma8.py."""
    from yolopoint_amd.plan import PlanBuilder
    monkeypatch.setattr(PlanBuilder, "autotune", False)


# Collection order of the GPU suite (`pytest -m gpu -x`): the oracle / golden-fixture parity files first, in the order of SURVEY.md section 8's
# rows; tuning and multi-process files next; statistical / convergence tests (training curves, trained-checkpoint parity) last, so that a
# noise-sensitive bar can never stand in front of the parity suite.  Within a file the order of definition is kept, except for tests that
# carry the `statistical` marker, which move to the very end of the whole run.
_FILE_ORDER = (
    "test_oracle_golden", "test_losses_golden", "test_backward_golden", "test_eval_oracle_golden", "test_host_layout", "test_config0_cpu_plumbing",
    "test_gpu_bench_shapes", "test_gpu_blocks", "test_gpu_model", "test_gpu_postproc", "test_gpu_losses_golden", "test_gpu_training",
    "test_gpu_wgrad", "test_gpu_sampling", "test_gpu_frontend", "test_gpu_export", "test_gpu_conv_tiles", "test_gpu_conv_mma8", "test_gpu_bn",
    "test_gpu_fp8", "test_dp_gloo", "test_gpu_dp_tuning", "test_gpu_accuracy_parity",
)


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}

    def key(pair):
        idx, item = pair
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return (1 if item.get_closest_marker("statistical") else 0, rank.get(stem, len(rank)), idx)
    items[:] = [item for _, item in sorted(enumerate(items), key=key)]
