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


def test_noalloc_allocates_nothing_and_the_allocating_entry_point_does(make_nets):
    """The property in the name, on a FRESH context (its pair buffer still empty): free device memory (hipMemGetInfo) is
    unchanged across rt_is_estimate_dev_noalloc and smaller after the first rt_is_estimate_dev of the same size (10 MB of
    pairs); rt_get_counters sees no launch of the no-alloc call (it makes no event pair), one of the allocating call."""
    import os
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    gw, gb, hw, hb = make_nets(2, 2)
    eng = _lib.Engine(0)
    try:
        eng.set_mlp(G, gw, gb)
        eng.set_mlp(H, hw, hb)
        N, S = 4000, 10000                                  # 157 pairs per point: 10 MB
        rng = np.random.RandomState(21)
        y = rng.standard_normal((N, 2)) * 0.5
        xhat = eng.mlp_forward(H, y.astype(np.float32))
        y_d, xh_d = eng.to_device(y), eng.to_device(xhat)
        o1, o2 = eng.to_device(np.zeros(N)), eng.to_device(np.zeros(N))
        eng.sync()
        eng.reset_counters()
        free0 = eng.dev_mem_info()[0]
        eng.is_estimate_dev_noalloc(y_d, xh_d, N, S, 0.1, 0.5, o2)
        eng.sync()
        free1 = eng.dev_mem_info()[0]
        assert free1 == free0, (free0, free1)
        assert eng.counters()["launches"] == 0
        eng.is_estimate_dev(y_d, xh_d, N, S, 0.1, 0.5, o1)
        eng.sync()
        free2 = eng.dev_mem_info()[0]
        assert free2 <= free1 - N * 157 * 16, (free1, free2)
        assert eng.counters()["launches"] == 1
        assert np.array_equal(eng.from_device(o1, (N,), np.float64), eng.from_device(o2, (N,), np.float64))
        for p in (y_d, xh_d, o1, o2):
            eng.dev_free(p)
    finally:
        eng.close()


def test_noalloc_is_refused_while_another_context_uses_the_scratch_array(make_nets):
    """All contexts on a GPU share the array: a second context's call while the first one's is outstanding gets RT_EBUSY
    (not a silent race); once the first has finished, the second is served, with the right bits."""
    import os
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    gw, gb, hw, hb = make_nets(2, 2)
    a, b = _lib.Engine(0), _lib.Engine(0)
    try:
        for e in (a, b):
            e.set_mlp(G, gw, gb)
            e.set_mlp(H, hw, hb)
        N, S = 2000, 20000                                   # 4e7 sample-evals: more than a second of GPU time
        y = np.random.RandomState(22).standard_normal((N, 2)) * 0.5
        xhat = a.mlp_forward(H, y.astype(np.float32))
        ya, xa, oa = a.to_device(y), a.to_device(xhat), a.to_device(np.zeros(N))
        yb, xb, ob = b.to_device(y[:50]), b.to_device(xhat[:50]), b.to_device(np.zeros(50))
        a.is_estimate_dev_noalloc(ya, xa, N, S, 0.1, 0.5, oa)            # returns at once; the kernel runs for a second
        with pytest.raises(_lib.RtError) as ei:
            b.is_estimate_dev_noalloc(yb, xb, 50, 640, 0.1, 0.5, ob)
        assert ei.value.code == _lib.RT_EBUSY
        a.is_estimate_dev_noalloc(ya, xa, 10, 640, 0.1, 0.5, oa)         # the owner itself: ordered by its stream
        a.sync()
        b.is_estimate_dev_noalloc(yb, xb, 50, 640, 0.1, 0.5, ob)         # the array is free again
        b.sync()
        assert np.array_equal(b.from_device(ob, (50,), np.float64), b.is_estimate(y[:50], xhat[:50], 640, 0.1, 0.5))
        for e, ps in ((a, (ya, xa, oa)), (b, (yb, xb, ob))):
            for p in ps:
                e.dev_free(p)
    finally:
        a.close()
        b.close()


def test_noalloc_through_the_share_measuring_first_block(tmp_path):
    """A fresh process's first long call measures the XCD shares on a first block of its query points (rt_is.hip,
    is_enqueue): through the no-alloc entry point that block and the rest both write the scratch array, in stream order.
    Same bits as a context with the balancing off; the shares have moved afterwards."""
    import os, subprocess, sys, textwrap
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text("import os, sys\nsys.path.insert(0, %r)\nsys.path.insert(0, %r)\n" % (root, os.path.join(root, "tests"))
                      + textwrap.dedent("""
        import numpy as np
        from roundtrip_amd import _lib
        from conftest import glorot
        gw, gb = glorot([2] + [512] * 10 + [2], 1234, 5)
        N, S = 2048, 19200                                   # 300 passes per point: 9.8 MB of pairs; first block = 274 points
        y = np.random.RandomState(3).standard_normal((N, 2)) * 0.5
        xhat = (y * 0.3).astype(np.float32)
        eng = _lib.Engine(0)
        eng.set_mlp(0, gw, gb)
        assert np.array_equal(eng.xcd_shares(), np.full(8, 0.125))
        y_d, xh_d, o_d = eng.to_device(y), eng.to_device(xhat), eng.to_device(np.zeros(N))
        eng.is_estimate_dev_noalloc(y_d, xh_d, N, S, 0.1, 0.5, o_d)
        eng.sync()
        got = eng.from_device(o_d, (N,), np.float64)
        assert not np.array_equal(eng.xcd_shares(), np.full(8, 0.125))          # measured on the first block
        os.environ["RT_IS_BALANCE"] = "0"
        plain = _lib.Engine(0)
        plain.set_mlp(0, gw, gb)
        assert np.array_equal(got, plain.is_estimate(y, xhat, S, 0.1, 0.5))
        print("noalloc first block ok")
    """))
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=280, cwd=str(tmp_path),
                         env={k: v for k, v in os.environ.items() if k not in ("RT_IS_BALANCE", "RT_IS_FIRST_BLOCK")})
    assert res.returncode == 0 and "noalloc first block ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
