import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_groups():
    return sorted(f[len("ref_state_"):-len(".npz")] for f in os.listdir(GOLDEN_DIR)
                  if f.startswith("ref_state_") and f.endswith(".npz"))


def cpu_share():
    """Cores this process may really use: the cgroup CPU quota if there is one, else the scheduler affinity.  (A GPU box hands a
    job a share of a large host: OpenMP's default - one thread per hardware thread of the HOST - oversubscribes that share
    many times over, and every small parallel region of the oracle then costs milliseconds.)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # before libgomp initialises: idle oracle threads sleep instead of spinning


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): compiled on first use with gcc; OpenMP threads = this process's CPU share."""
    from oracle import binding
    binding.lib().tso_set_num_threads(min(cpu_share(), 32))
    return binding


def load_golden(name):
    import numpy as np
    with np.load(os.path.join(GOLDEN_DIR, f"ref_state_{name}.npz")) as z:
        return {k: z[k] for k in z.files}
