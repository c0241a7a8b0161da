"""`bench.py --gpus N` starts N ranks (CPU, gloo, no GPU work).

The reference is started one process per GPU (`accelerate launch src/train.py`, README.md:76; train.py:38-46,174).  bench.py accepts both
forms: the driver's (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) and the bare one (`python bench.py --gpus N`,
which re-executes itself under torch.distributed.run).  `--backend gloo --dry` runs the launcher + process group + the timed-region protocol
of the real records with a stub step, so the rank plumbing is checked here without a GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"the contract is ONE JSON line on stdout, got {len(lines)}: {stdout[:400]}"
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bare_invocation_self_launches_n_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--backend", "gloo", "--dry", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_line(r.stdout)
    assert rec["n_gpus"] == n and rec["dry"] is True
    assert rec["config"]["ranks_reported"] == list(range(n))
    assert len({x["pid"] for x in rec["ranks"]}) == n, "N distinct processes"
    assert [x["local_rank"] for x in rec["ranks"]] == list(range(n))
    assert rec["config"]["allreduce_sum"] == n * (n + 1) / 2, "every rank took part in the collective of the stub step"
    assert "torch.distributed.run" in r.stderr and f"--nproc-per-node={n}" in r.stderr


def test_driver_form_under_torch_distributed_run():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           BENCH, "--gpus", "2", "--backend", "gloo", "--dry", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_reported"] == [0, 1]


def test_more_gpus_than_the_box_has_fails_loudly():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 1 if have else 2)], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert r.stdout.strip() == "", "no record may be printed for a run that did not start N ranks"
    assert "visible GPUs" in r.stderr


def test_world_size_and_gpus_must_agree():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""
