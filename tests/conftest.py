import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask AND cgroup quota (a GPU box shows 256 CPUs in the affinity
    mask of a 16-core container; 256 oracle threads on 16 cores run the CPU checker ~40x slower)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests (tests/test_sharding_gpu.py) start their rank processes from a fork server that is
    # launched HERE, before anything in this process initialises HIP: a process that has touched the GPU must not exec
    # another program, and a fork server forks clean children without any exec.
    try:
        from multiprocessing import forkserver

        forkserver.ensure_running()
    except Exception:  # noqa: BLE001
        pass
    try:
        import torch

        torch.set_num_threads(host_cores())  # the CPU oracle is the checker in most tests
    except Exception:  # noqa: BLE001
        pass


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with `-m gpu`; without a GPU they are skipped, never faked.
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
