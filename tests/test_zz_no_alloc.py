"""rt_is_estimate_dev_noalloc (include/rt_is.h): the entry point that allocates nothing — its per-pass pairs go to the
per-GPU `scratch` array — against rt_is_estimate_dev, bit for bit, through the C ABI.  Collected last (file name): the
round's end runs pytest -x."""
import numpy as np
import pytest

from roundtrip_amd import _lib

pytestmark = pytest.mark.gpu

G, H = _lib.RT_MLP_G, _lib.RT_MLP_H


def _both(engine, y, xhat, S, sd, scale, log, base, **kw):
    """(rt_is_estimate_dev, rt_is_estimate_dev_noalloc) results for the same device-resident inputs."""
    N = y.shape[0]
    y_d, xh_d = engine.to_device(y), engine.to_device(xhat)
    o1, o2 = engine.to_device(np.full(N, -7.0)), engine.to_device(np.full(N, -9.0))
    try:
        engine.is_estimate_dev(y_d, xh_d, N, S, sd, scale, o1, log=log, y_index_base=base, **kw)
        engine.is_estimate_dev_noalloc(y_d, xh_d, N, S, sd, scale, o2, log=log, y_index_base=base, **kw)
        engine.sync()
        return engine.from_device(o1, (N,), np.float64), engine.from_device(o2, (N,), np.float64)
    finally:
        for p in (y_d, xh_d, o1, o2):
            engine.dev_free(p)


def test_noalloc_matches_the_allocating_entry_point_twice_in_a_row(engine, make_nets):
    gw, gb, hw, hb = make_nets(2, 2)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    rng = np.random.RandomState(11)
    # first call: 300 points x S = 1000 (15 full passes + a packed leftover entry per point)
    y = rng.standard_normal((300, 2)) * 0.5
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    a, b = _both(engine, y, xhat, 1000, 0.1, 0.5, True, 0)
    assert np.all(np.isfinite(a)) and np.array_equal(a, b)
    # second call, smaller (fewer points, fewer samples, another base): the array still holds the first call's pairs
    y2, xhat2 = y[40:77], xhat[40:77]
    a2, b2 = _both(engine, y2, xhat2, 200, 0.1, 0.5, False, 40, seed=5)
    assert np.all(np.isfinite(a2)) and np.array_equal(a2, b2)
    # and it is the same estimate the host-pointer entry gives
    np.testing.assert_array_equal(b2, engine.is_estimate(y2, xhat2, 200, 0.1, 0.5, log=False, seed=5, y_index_base=40))


def test_noalloc_on_another_shape_and_the_size_limit(engine, make_nets):
    gw, gb, hw, hb = make_nets(20, 90)                    # the SHARE sampler instantiation, 3 output tiles
    engine.set_mlp(G, gw, gb)
    rng = np.random.RandomState(12)
    y = rng.standard_normal((9, 90)) * 0.3
    xhat = (rng.standard_normal((9, 20)) * 0.3).astype(np.float32)
    a, b = _both(engine, y, xhat, 777, 0.5, 0.1, True, 123456789012)
    assert np.array_equal(a, b)
    # pairs beyond the 32 MiB array: refused with RT_ETOOBIG before anything is enqueued; the message names the limit
    S = 64
    n_max = _lib.RT_NOALLOC_SCRATCH_BYTES // 16           # one pair per point at S = 64
    n = n_max + 1
    y_d = engine.dev_alloc(n * 90 * 8)
    xh_d = engine.dev_alloc(n * 20 * 4)
    o_d = engine.dev_alloc(n * 8)
    try:
        with pytest.raises(_lib.RtError) as ei:
            engine.is_estimate_dev_noalloc(y_d, xh_d, n, S, 0.5, 0.1, o_d)
        assert ei.value.code == _lib.RT_ETOOBIG and str(n_max) in str(ei.value)
    finally:
        for p in (y_d, xh_d, o_d):
            engine.dev_free(p)
