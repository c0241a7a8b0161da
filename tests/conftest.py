import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


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
