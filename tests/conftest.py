import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_tables(g, prefix="t_"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def gpu_ctx():
    from pythtb_amd import _lib
    return _lib.default_context()
