import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "reinmav-gym_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    out = {}
    for kind in ("quad2d", "quad2d_sl", "quad3d", "quad3d_sl"):
        out[kind] = dict(np.load(os.path.join(ROOT, "tests", "golden", kind + ".npz")))
    return out


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the HIP library, the oracle and the host-math helper."""
    import subprocess

    for d in ("reinmav-gym_amd", "oracle", os.path.join("tests", "hostmath")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, d)], check=True)
    return True
