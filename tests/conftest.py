import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with `-m gpu` on the GPU box)")


def _gpu_box():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def engine():
    """One rt_ctx on HIP device 0 for the whole session.  On a machine without a GPU the test is skipped;
    on a GPU box a failure to create it is an error, never a skip."""
    from roundtrip_amd import _lib
    if not _gpu_box():
        pytest.skip("no GPU on this machine")
    eng = _lib.Engine(0)
    yield eng
    eng.close()


def glorot(dims, seed=1234, bias_seed=None):
    from oracle import is_ref
    ws, bs = is_ref.glorot_mlp(dims, np.random.RandomState(seed))
    if bias_seed is not None:
        r = np.random.RandomState(bias_seed)
        bs = [r.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in bs]
    return ws, bs


@pytest.fixture(scope="session")
def make_nets():
    """make_nets(dx, dy, g_hidden=512, h_hidden=256, n_hidden=10) -> (gw, gb, hw, hb), cached."""
    cache = {}

    def make(dx, dy, g_hidden=512, h_hidden=256, n_hidden=10, biased=True):
        key = (dx, dy, g_hidden, h_hidden, n_hidden, biased)
        if key not in cache:
            gw, gb = glorot([dx] + [g_hidden] * n_hidden + [dy], 1234, 5 if biased else None)
            hw, hb = glorot([dy] + [h_hidden] * n_hidden + [dx], 4321, 6 if biased else None)
            cache[key] = (gw, gb, hw, hb)
        return cache[key]

    return make
