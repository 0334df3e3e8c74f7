import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the kernel-choice overrides the tests use (UGSM_MARCH_MIN_PIXELS, UGSM_SMALL_RH, ...) are development switches: the library
# reads them only under UGSM_DEV=1 (ugsm_runtime.cpp, apply_dev_env)
os.environ["UGSM_DEV"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.argwhere(~same)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ; first at {tuple(bad[0])}: "
                             f"{a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}; max abs diff {np.nanmax(d):.3e}")
