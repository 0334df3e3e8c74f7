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


def _oracle_16mp(orc, seed_off, with_stack):
    from ug_stereomatcher_amd import synth
    W, H = 4928, 3264
    L, R, dx, dy = synth.make_pair(W, H, synth.BASE_SEED + seed_off)
    orc.set_num_threads(16)
    try:
        full = orc.match_full(L, R, 14)
        stack = orc.match_foveated(L, R, 14, 7)[0] if with_stack else None   # (3, F, fovH, fovW)
    finally:
        orc.set_num_threads(8)
    return dict(W=W, H=H, L=L, R=R, dx=dx, dy=dy, full=full, stack=stack)


@pytest.fixture(scope="session")
def oracle_16mp(orc):
    """The 16 MP synthetic pair of BASELINE configs[2] / configs[3] (bench.py's first pair) and the live oracle's answers for it (full
    pyramid: about 3 s on the GPU box's 16 threads; foveated stack: under a second), shared by every full-size parity test."""
    return _oracle_16mp(orc, 2, True)


@pytest.fixture(scope="session")
def oracle_16mp_b(orc):
    """bench.py's second 16 MP pair (seed + 16) and the oracle's full-mode answer for it."""
    return _oracle_16mp(orc, 2 + 16, False)


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
