"""bench.py --gpus N started WITHOUT torch.distributed.run must start its N ranks itself (as a child process, before anything
touches the GPU) and relay rank 0's JSON line: a driver that runs `python bench.py --gpus 8` gets an 8-rank measurement."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def run_bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(REPO / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(REPO))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}: {p.stdout[-500:]}"
    return json.loads(lines[0])


def test_self_launch_starts_two_ranks_dry_run():
    """No GPU needed: rendezvous on 127.0.0.1 over gloo, one all-reduce, rank 0's line."""
    r = run_bench(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert r["n_gpus"] == 2 and r["dry_run"] is True


def test_single_rank_dry_run_does_not_spawn():
    r = run_bench(["--gpus", "1", "--dry-run"])
    assert r["n_gpus"] == 1


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu():
    """Two ranks sharing the one GPU of the box (MZ_BENCH_BACKEND=gloo): the real step incl. the overlapped gather of outputs."""
    r = run_bench(["--gpus", "2", "--workload", "cfg2", "--images-per-gpu", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                   "--no-microbench"], {"MZ_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["value"] > 0
    assert r["config"]["global_batch"] == 4
    assert r["value_without_gather"] is not None and r["value_without_gather"] > 0
    assert "bf16" in r["config"]["workload"]
