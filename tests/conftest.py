import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _limit_cpu_threads()


def _limit_cpu_threads():
    """The oracle (torch on the CPU) runs inside the test process.  A one-GPU box shows all of the host's cores but grants 16: with
    torch's default -- one thread per visible core -- the oracle's full-size runs spend their time being throttled (177 s for one
    forward + backward at 1 x 8192 points that takes 5 s on 8 unshared cores).  Same rule as bench.py's usable_cores()."""
    import torch
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    n = int(os.environ.get("FSG_CPU_THREADS", max(1, min(n, 16))))
    torch.set_num_threads(n)
    os.environ.setdefault("OMP_NUM_THREADS", str(n))


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
