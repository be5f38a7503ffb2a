import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-integration_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _usable_cores() -> int:
    """Cores this process may really use (affinity mask and cgroup quota): the GPU box shows 128 logical CPUs to a 16-core share, and
    the CPU oracle on 128 threads runs at half the speed it reaches on 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    import torch
    n = min(_usable_cores(), 32)
    torch.set_num_threads(n)
    os.environ.setdefault("OMP_NUM_THREADS", str(n))  # explicit: Trainer.setup() (ssi.train_utils.limit_host_threads) then leaves the oracle's threads alone
    yield


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
