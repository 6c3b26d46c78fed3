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


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): compiled on first use with gcc."""
    from oracle import binding
    binding.lib()
    return binding


def load_golden(name):
    import numpy as np
    with np.load(os.path.join(GOLDEN_DIR, f"ref_state_{name}.npz")) as z:
        return {k: z[k] for k in z.files}
