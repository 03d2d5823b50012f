"""GPU parity tests: the HIP path, called through the C ABI (roundtrip_amd._lib.Engine is a thin ctypes
layer over include/rt_is.h), against the CPU oracle on the same seeded inputs, against the committed golden
vectors, and through size-independent properties at BASELINE sizes.

Tolerances (north_star: 1e-4 relative on log p(y)):
  * MLP forward vs the oracle run in the kernel's summation order: BIT-EXACT (the fp32 MFMA is a k-ordered
    fmaf chain);
  * log p(y) vs the oracle in kernel order: 1e-12 relative (only fp64 rounding of the scoring differs);
  * log p(y) vs the oracle in natural order / the numpy golden vectors: 1e-6 relative.
"""
import os

import numpy as np
import pytest
from scipy import stats

from oracle import c_oracle, is_ref
from roundtrip_amd import _lib

from test_oracle import GOLDEN, GOLDEN_NAMES, assert_linear_close, gaussian_logpdf, golden_nets, golden_samples, linear_g

pytestmark = pytest.mark.gpu

G, H = _lib.RT_MLP_G, _lib.RT_MLP_H

# (dx, dy) of the BASELINE configs: 2/2, AReM 3/6, CASP 5/9, BANK 8/17, HEPMASS 8/21, YearPredictionMSD 20/90
DIMS = [(2, 2), (3, 6), (5, 9), (8, 17), (8, 21), (20, 90)]


def glorot_nets(dx, hid, dy, seed, n_hidden=4):
    """A biased Glorot generator [dx, hid x n_hidden, dy]."""
    rng = np.random.RandomState(seed)
    ws, bs = is_ref.glorot_mlp([dx] + [hid] * n_hidden + [dy], rng)
    return ws, [rng.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in bs]


def reference_samples(xhat, scale, df, S, seed=0):
    """Draws as the reference does (main_density_est.py:316-318) with numpy's legacy global RNG."""
    np.random.seed(seed)
    return np.stack([is_ref.sample_from_qx(xhat[n].astype(np.float64), xhat.shape[1], scale, df, S)
                     for n in range(xhat.shape[0])])


@pytest.mark.parametrize("dx,dy", DIMS)
def test_mlp_forward_bit_exact(engine, make_nets, dx, dy):
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    og, oh = c_oracle.OracleMlp(gw, gb), c_oracle.OracleMlp(hw, hb)
    rng = np.random.RandomState(dx * 100 + dy)
    for n in (1, 63, 64, 65, 200):                      # ragged row counts around the 64-row pass
        x = (rng.standard_normal((n, dx)) * 2).astype(np.float32)
        got = engine.mlp_forward(G, x)
        assert np.array_equal(got, og.forward(x, c_oracle.ORDER_KERNEL, nthreads=8))
        nat = og.forward(x, c_oracle.ORDER_NATURAL, nthreads=8)
        assert np.abs(got - nat).max() <= 4e-6 * np.abs(nat).max()
    y = rng.standard_normal((130, dy)).astype(np.float32)
    assert np.array_equal(engine.mlp_forward(H, y), oh.forward(y, c_oracle.ORDER_KERNEL, nthreads=8))
    assert engine.mlp_forward(G, np.zeros((0, dx), np.float32)).shape == (0, dy)


def test_mlp_forward_heavy_tails_and_specials(engine, make_nets):
    """Cauchy proposals routinely give |x| ~ 1e4..1e6; inf/NaN inputs must propagate, not crash."""
    gw, gb, _, _ = make_nets(2, 2)
    engine.set_mlp(G, gw, gb)
    og = c_oracle.OracleMlp(gw, gb)
    x = np.array([[1e4, -3e5], [1e8, 1e-8], [0, 0], [-0.0, 1e-30], [1e30, -1e30], [np.inf, 1.0], [np.nan, 0.0]],
                 np.float32)
    got = engine.mlp_forward(G, x)
    want = og.forward(x, c_oracle.ORDER_KERNEL)
    assert np.array_equal(got[:5], want[:5])
    assert not np.isfinite(got[5]).any() and np.isnan(got[6]).all()


@pytest.mark.parametrize("hid,nl,din", [(128, 1, 3), (128, 3, 3), (256, 2, 3), (512, 4, 3),
                                        # one hidden layer; input dims that take 2, 3 and 5 k-groups in layer 0
                                        (512, 1, 12), (256, 1, 8), (256, 3, 16), (512, 2, 40), (128, 2, 20)])
def test_other_depths_and_widths(engine, hid, nl, din):
    from conftest import glorot
    ws, bs = glorot([din] + [hid] * nl + [5], 9, 10)
    engine.set_mlp(G, ws, bs)
    x = np.random.RandomState(1).standard_normal((100, din)).astype(np.float32)
    assert np.array_equal(engine.mlp_forward(G, x), c_oracle.OracleMlp(ws, bs).forward(x, c_oracle.ORDER_KERNEL))


def _pad_hidden(ws, bs, hid):
    """Zero-pads every hidden layer to width hid (what rt_set_mlp does internally)."""
    pw, pb = [], []
    for l, (w, b) in enumerate(zip(ws, bs)):
        kp = w.shape[0] if l == 0 else hid
        nfp = w.shape[1] if l == len(ws) - 1 else hid
        W = np.zeros((kp, nfp), np.float32)
        W[:w.shape[0], :w.shape[1]] = w
        B = np.zeros(nfp, np.float32)
        B[:b.shape[0]] = b
        pw.append(W)
        pb.append(B)
    return pw, pb


@pytest.mark.parametrize("dims,hid", [([3, 100, 100, 5], 128), ([4, 64, 200, 32, 7], 256), ([2, 300, 512, 9], 512),
                                      ([3, 1024, 1024, 1024, 6], 1024), ([5, 700, 600, 3], 1024)])
def test_any_hidden_widths_by_exact_padding(engine, dims, hid):
    """v2 accepts any g_units list (src/pyroundtrip/model.py:7,17-18): hidden layers are zero-padded to the
    kernel width (128/256/512/1024).  Bit-exact against the oracle on explicitly padded weights, close to the
    plain (unpadded, natural order) oracle; width 1024 runs the 32-sample-per-pass kernel."""
    from conftest import glorot
    ws, bs = glorot(dims, 21, 22)
    engine.set_mlp(G, ws, bs)
    x = np.random.RandomState(2).standard_normal((131, dims[0])).astype(np.float32)
    got = engine.mlp_forward(G, x)
    pw, pb = _pad_hidden(ws, bs, hid)
    assert np.array_equal(got, c_oracle.OracleMlp(pw, pb).forward(x, c_oracle.ORDER_KERNEL, nthreads=8))
    nat = c_oracle.OracleMlp(ws, bs).forward(x, c_oracle.ORDER_NATURAL, nthreads=8)
    assert np.abs(got - nat).max() <= 3e-6 * np.abs(nat).max()
    # and through the estimator
    hw, hb = glorot([dims[-1], 128, dims[0]], 23, 24)
    engine.set_mlp(H, hw, hb)
    y = np.random.RandomState(3).standard_normal((3, dims[-1])) * 0.2
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    S = 200
    est = engine.is_estimate(y, xhat, S, 0.3, 0.2, 1.0, seed=5)
    xs = engine.draw_samples(xhat, S, 0.2, 1.0, seed=5).astype(np.float64)
    want = c_oracle.is_estimate_given_samples(c_oracle.OracleMlp(pw, pb), y, xhat, xs, 0.3, 0.2, 1.0, True,
                                              c_oracle.ORDER_KERNEL, 8)
    np.testing.assert_allclose(est, want, rtol=1e-12)
    c = engine.counters()
    assert c["flops"] > 0


@pytest.mark.parametrize("dx,dy", DIMS)
@pytest.mark.parametrize("log", [True, False])
def test_given_samples_match_oracle(engine, make_nets, dx, dy, log):
    """The reference's own draws (numpy legacy RNG, float64) scored on the GPU and by the oracle."""
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    og = c_oracle.OracleMlp(gw, gb)
    y = np.random.RandomState(42).standard_normal((5, dy)) * 0.3
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    S = 777                                            # not a multiple of 64: ragged last pass
    xs = reference_samples(xhat, 0.1, 1, S)
    got = engine.is_estimate_given_samples(y, xhat, xs, 0.1, 0.1, 1.0, log=log)
    exact = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.1, 0.1, 1.0, log, c_oracle.ORDER_KERNEL, 8)
    nat = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.1, 0.1, 1.0, log, c_oracle.ORDER_NATURAL, 8)
    if log:
        np.testing.assert_allclose(got, exact, rtol=1e-12)
        np.testing.assert_allclose(got, nat, rtol=1e-6)
    else:
        assert_linear_close(got, exact, 1e-12)
        assert_linear_close(got, nat, 1e-6)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_golden_vectors(engine, name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    gw, gb, hw, hb = golden_nets(f)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    xhat = engine.mlp_forward(H, f["y"].astype(np.float32))
    np.testing.assert_allclose(xhat, f["xhat"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(engine.mlp_forward(G, f["xhat"]), f["g_of_xhat"], rtol=2e-5, atol=1e-7)
    xs = golden_samples(f)
    sd, scale, df = float(f["sd"]), float(f["scale"]), float(f["df"])
    got = engine.is_estimate_given_samples(f["y"], f["xhat"], xs, sd, scale, df, log=True)
    np.testing.assert_allclose(got, f["log_est"], rtol=1e-6)
    lin = engine.is_estimate_given_samples(f["y"], f["xhat"], xs, sd, scale, df, log=False)
    assert_linear_close(lin, f["lin_est"], 1e-6)


@pytest.mark.parametrize("dx,dy,df", [(2, 2, 1.0), (3, 6, 1.0), (20, 90, 1.0), (2, 2, 3.0), (5, 9, 2.5)])
def test_rng_mode_matches_oracle_on_the_same_draws(engine, make_nets, dx, dy, df):
    """rt_is_estimate (on-GPU Philox draws) == oracle scoring of rt_draw_samples' output."""
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    og = c_oracle.OracleMlp(gw, gb)
    y = np.random.RandomState(3).standard_normal((6, dy)) * 0.3
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    S = 1500
    got = engine.is_estimate(y, xhat, S, 0.1, 0.1, df, seed=7, y_index_base=100)
    xs = engine.draw_samples(xhat, S, 0.1, df, seed=7, y_index_base=100)
    want = c_oracle.is_estimate_given_samples(og, y, xhat, xs.astype(np.float64), 0.1, 0.1, df, True,
                                              c_oracle.ORDER_KERNEL, 8)
    np.testing.assert_allclose(got, want, rtol=1e-12)
    # the oracle's restatement of the sampler agrees to a few float32 ulp (libm vs device logf/sincosf); on the gamma
    # path (non-integer df) a borderline accept/reject may fall the other way once in a great while
    xo = c_oracle.draw_samples(xhat, 100, S, 0.1, df, 7)
    close = np.isclose(xs, xo, rtol=2e-3, atol=1e-6)
    assert close.mean() > (0.9999 if df == float(int(df)) else 0.999)


def test_sampler_statistics(engine, make_nets):
    """df = 1: each coordinate of (x - xhat)/scale is standard Cauchy, the radius has the multivariate-Cauchy
    law; df = 3, 2.5: Student-t marginals (the 2.5 case exercises the gamma sampler)."""
    gw, gb, _, _ = make_nets(2, 2)
    engine.set_mlp(G, gw, gb)
    xhat = np.array([[0.5, -1.0]], np.float32)
    n = 200000
    for df in (1.0, 3.0, 2.5):
        xs = engine.draw_samples(xhat, n, 0.5, df, seed=0)[0].astype(np.float64)
        u = (xs - xhat[0]) / 0.5
        for d in range(2):
            ks = stats.kstest(u[:, d], stats.t(df).cdf)
            assert ks.pvalue > 1e-3, (df, d, ks)
        # radial law of a bivariate t: r^2/2 ~ F(2, df)
        ks = stats.kstest((u ** 2).sum(axis=1) / 2.0, stats.f(2, df).cdf)
        assert ks.pvalue > 1e-3, (df, ks)
    a = engine.draw_samples(xhat, 1000, 0.5, 1.0, seed=0)
    assert np.array_equal(a, engine.draw_samples(xhat, 1000, 0.5, 1.0, seed=0))
    assert not np.array_equal(a, engine.draw_samples(xhat, 1000, 0.5, 1.0, seed=1))
    assert not np.array_equal(a, engine.draw_samples(xhat, 1000, 0.5, 1.0, seed=0, y_index_base=1))


def test_sharding_and_chunking_invariance(engine, make_nets):
    """Results are a function of (weights, y, xhat, S, seed, GLOBAL query index) only: every pass reduces its samples of
    one query point to a (max, sum) pair and a point's pairs are merged in a fixed order, so neither the number of query
    points in a call, nor a rank's offset, nor how passes were dealt to workgroups changes a single bit — what makes any
    GPU count give the same answer.  S = 650 = 10 full passes + 10 samples: exercises the packed leftover passes too."""
    gw, gb, hw, hb = make_nets(2, 2)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    y = np.random.RandomState(5).standard_normal((601, 2)) * 0.2
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    for S in (640, 650, 37):
        full = engine.is_estimate(y, xhat, S, 0.05, 0.5, 1.0, seed=0)
        assert np.array_equal(full, engine.is_estimate(y, xhat, S, 0.05, 0.5, 1.0, seed=0))     # deterministic
        parts = [engine.is_estimate(y[lo:hi], xhat[lo:hi], S, 0.05, 0.5, 1.0, seed=0, y_index_base=lo)
                 for lo, hi in ((0, 300), (300, 305), (305, 601))]
        assert np.array_equal(np.concatenate(parts), full)
        one = engine.is_estimate(y[7:8], xhat[7:8], S, 0.05, 0.5, 1.0, seed=0, y_index_base=7)   # one point alone
        assert np.array_equal(one, full[7:8])


def test_large_calls_split_into_bounded_launches(engine, make_nets, monkeypatch):
    """A call above the per-launch bound (2^28 sample-evals by default; RT_IS_MAX_LAUNCH_EVALS, read when a context is
    created) runs as several launches over blocks of query points.  The split changes no draw and no result: plain,
    several-sd and given-samples calls agree with the one-launch answers."""
    gw, gb, hw, hb = make_nets(3, 6)
    y = np.random.RandomState(15).standard_normal((37, 6)) * 0.1
    S = 700
    monkeypatch.setenv("RT_IS_MAX_LAUNCH_EVALS", str(5 * S + 1))            # 5 points per launch: 8 launches, ragged tail
    small = _lib.Engine(0)
    try:
        for e in (engine, small):
            e.set_mlp(G, gw, gb)
            e.set_mlp(H, hw, hb)
        xhat = engine.mlp_forward(H, y.astype(np.float32))
        engine.reset_counters(); small.reset_counters()
        one = engine.is_estimate(y, xhat, S, 0.3, 0.1, 1.0, seed=3, y_index_base=11)
        many = small.is_estimate(y, xhat, S, 0.3, 0.1, 1.0, seed=3, y_index_base=11)
        assert engine.counters()["launches"] == 1 and small.counters()["launches"] == 8
        assert small.counters()["sample_evals"] == 37 * S
        assert np.array_equal(many, one)                       # bit for bit: the split changes no draw and no merge order
        sds = [0.05, 0.3, 1.0]
        assert np.array_equal(small.is_estimate_multi(y, xhat, S, sds, 0.1, 1.0, seed=3),
                              engine.is_estimate_multi(y, xhat, S, sds, 0.1, 1.0, seed=3))
        E = engine.is_pair_count(S)                          # pair ranges are computed in blocks of query points too
        assert np.array_equal(small.is_pairs(y, xhat, S, sds, 0.1, 2, E, seed=3),
                              engine.is_pairs(y, xhat, S, sds, 0.1, 2, E, seed=3))
        xs = engine.draw_samples(xhat, S, 0.1, 1.0, seed=3).astype(np.float64)
        assert np.array_equal(small.is_estimate_given_samples(y, xhat, xs, 0.3, 0.1, 1.0, log=False),
                              engine.is_estimate_given_samples(y, xhat, xs, 0.3, 0.1, 1.0, log=False))
    finally:
        small.close()


def test_event_pool_is_bounded_over_many_launches(engine, make_nets, monkeypatch):
    """The per-launch HIP events that feed rt_get_counters are harvested every 512 launches, so a caller that never
    reads the counters holds a bounded number of them; the counters still cover every launch."""
    gw, gb, hw, hb = make_nets(2, 2)
    y = np.random.RandomState(16).standard_normal((1100, 2)) * 0.2
    S = 64
    monkeypatch.setenv("RT_IS_MAX_LAUNCH_EVALS", str(S))                     # one query point per launch
    small = _lib.Engine(0)
    try:
        for e in (engine, small):
            e.set_mlp(G, gw, gb)
            e.set_mlp(H, hw, hb)
        xhat = engine.mlp_forward(H, y.astype(np.float32))
        small.reset_counters()
        many = small.is_estimate(y, xhat, S, 0.1, 0.5, 1.0, seed=2)
        c = small.counters()
        assert c["launches"] == 1100 and c["sample_evals"] == 1100 * S and c["kernel_ms"] > 0
        assert np.array_equal(many, engine.is_estimate(y, xhat, S, 0.1, 0.5, 1.0, seed=2))
    finally:
        small.close()


@pytest.mark.parametrize("S", [1, 2, 16, 17, 31, 32, 33, 63, 64, 65, 80, 96, 129])
def test_ragged_sample_counts(engine, make_nets, S):
    """S % 64 in (0, 32]: the last samples of 64 // (S % 64) query points share one leftover pass; above 32 a point's
    last pass is simply part full.  7 query points, so the last leftover pass is itself ragged.  Plain, several-sd and
    given-samples calls against the oracle on the same draws."""
    gw, gb, hw, hb = make_nets(3, 6)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    og = c_oracle.OracleMlp(gw, gb)
    y = np.random.RandomState(8).standard_normal((7, 6)) * 0.1
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    got = engine.is_estimate(y, xhat, S, 0.5, 0.1, 1.0, seed=1)
    xs = engine.draw_samples(xhat, S, 0.1, 1.0, seed=1).astype(np.float64)
    want = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.5, 0.1, 1.0, True, c_oracle.ORDER_KERNEL)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)   # atol: log p(y) itself can be ~1e-3 (M + log t - log S cancels)
    np.testing.assert_allclose(engine.is_estimate_given_samples(y, xhat, xs, 0.5, 0.1, 1.0), want, rtol=1e-12, atol=1e-14)   # atol: log p(y) itself can be ~1e-3 (M + log t - log S cancels)
    multi = engine.is_estimate_multi(y, xhat, S, [0.5, 0.2], 0.1, 1.0, seed=1)
    assert np.array_equal(multi[0], got)
    want2 = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.2, 0.1, 1.0, True, c_oracle.ORDER_KERNEL)
    np.testing.assert_allclose(multi[1], want2, rtol=1e-12, atol=1e-14)   # atol: log p(y) itself can be ~1e-3 (M + log t - log S cancels)


@pytest.mark.parametrize("S", [1, 8, 16, 17, 40, 100])
def test_ragged_sample_counts_on_32_sample_passes(engine, S):
    """Hidden width 1024 runs 32-sample passes: leftovers are packed for S % 32 in (0, 16]."""
    rng = np.random.RandomState(31)
    gw, gb = is_ref.glorot_mlp([3, 1024, 1024, 4], rng)
    for b in gb:
        b += rng.standard_normal(b.shape).astype(np.float32) * 0.05
    engine.set_mlp(G, gw, gb)
    og = c_oracle.OracleMlp(gw, gb)
    y = rng.standard_normal((5, 4)) * 0.1
    xhat = (rng.standard_normal((5, 3)) * 0.1).astype(np.float32)
    got = engine.is_estimate(y, xhat, S, 0.5, 0.2, 1.0, seed=4)
    xs = engine.draw_samples(xhat, S, 0.2, 1.0, seed=4).astype(np.float64)
    want = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.5, 0.2, 1.0, True, c_oracle.ORDER_KERNEL)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)   # atol: log p(y) itself can be ~1e-3 (M + log t - log S cancels)


def test_empty_and_error_paths(engine, make_nets):
    gw, gb, hw, hb = make_nets(2, 2)
    fresh = _lib.Engine(0)
    try:
        with pytest.raises(_lib.RtError):                       # MLP not set
            fresh.is_estimate(np.zeros((1, 2)), np.zeros((1, 2), np.float32), 10, 0.1, 0.1)
        with pytest.raises(_lib.RtError):                       # hidden width beyond the kernels' 1024
            fresh.set_mlp(G, [np.zeros((2, 1100), np.float32), np.zeros((1100, 2), np.float32)],
                          [np.zeros(1100, np.float32), np.zeros(2, np.float32)])
        with pytest.raises(_lib.RtError):                       # output dim > 128
            fresh.set_mlp(G, [np.zeros((2, 128), np.float32), np.zeros((128, 200), np.float32)],
                          [np.zeros(128, np.float32), np.zeros(200, np.float32)])
        with pytest.raises(_lib.RtError):                       # alpha outside [0, 1]
            fresh.set_mlp(G, gw, gb, alpha=1.5)
    finally:
        fresh.close()
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    assert engine.is_estimate(np.zeros((0, 2)), np.zeros((0, 2), np.float32), 100, 0.1, 0.1).shape == (0,)
    y, xh = np.zeros((2, 2)), np.zeros((2, 2), np.float32)
    for bad in (dict(sample_size=0), dict(sd_y=0.0), dict(scale=-1.0), dict(df=0.0)):
        kw = dict(sample_size=10, sd_y=0.1, scale=0.1, df=1.0)
        kw.update(bad)
        with pytest.raises(_lib.RtError):
            engine.is_estimate(y, xh, **kw)
    with pytest.raises(ValueError):
        engine.is_estimate(np.zeros((2, 3)), xh, 10, 0.1, 0.1)


def test_far_points_underflow_like_the_reference(engine, make_nets):
    """log=False far from the data: the reference's linear-space product underflows to exactly 0
    (main_density_est.py:290,344-345); log=True stays finite."""
    gw, gb, hw, hb = make_nets(2, 2)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    y = np.array([[300.0, -300.0], [0.01, 0.02]])
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    lin = engine.is_estimate(y, xhat, 500, 0.05, 0.5, 1.0, log=False)
    lg = engine.is_estimate(y, xhat, 500, 0.05, 0.5, 1.0, log=True)
    assert lin[0] == 0.0 and lin[1] > 0.0 and np.isfinite(lg).all() and lg[0] < -1e6
    np.testing.assert_allclose(np.log(lin[1]), lg[1], rtol=1e-12)


def test_analytic_linear_generator_known_answer_on_gpu(engine):
    """p(y) = N(y; b, A^T A + sd^2 I) for an exactly linear G (see tests/test_oracle.py); on-GPU draws, S large."""
    dx, dy, sd = 1, 3, 0.5
    A = np.array([[1.0, 0.5, -0.25]], np.float32)
    b = np.array([0.1, -0.2, 0.3], np.float32)
    ws, bs = linear_g(dx, dy, 128, 3, A, b)
    engine.set_mlp(G, ws, bs)
    cov = A.T.astype(np.float64) @ A + sd * sd * np.eye(dy)
    ys = np.array([[0.3, -0.1, 0.2], [1.0, 0.5, -0.5], [-0.4, 0.0, 0.9]])
    post_cov = np.linalg.inv(np.eye(dx) + A.astype(np.float64) @ A.T / sd ** 2)
    centres = np.stack([post_cov @ (A.astype(np.float64) @ (y - b)) / sd ** 2 for y in ys]).astype(np.float32)
    want = np.array([gaussian_logpdf(y, b.astype(np.float64), cov) for y in ys])
    for df in (1.0, 3.0):
        got = engine.is_estimate(ys, centres, 400000, sd, 1.0, df, seed=3)
        assert np.abs(got - want).max() < 0.01, (df, got, want)


def test_full_size_properties(engine, make_nets):
    """BASELINE config 2 at full size (100x100 grid, S = 10000, log=False): too large for the oracle, so
    check size-independent properties: exp(log-mode) == linear-mode, rank-sharding invariance on a slice,
    and oracle agreement on a few points re-evaluated alone."""
    gw, gb, hw, hb = make_nets(2, 2, biased=False)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    grid = is_ref.create_2d_grid_data(-5, 5, -5, 5, n=100)[2]
    xhat = engine.mlp_forward(H, grid.astype(np.float32))
    S = 10000
    lin = engine.is_estimate(grid, xhat, S, 0.1, 0.5, 1.0, log=False)
    lg = engine.is_estimate(grid, xhat, S, 0.1, 0.5, 1.0, log=True)
    assert lin.shape == (10000,) and np.isfinite(lg).all()
    assert_linear_close(lin, np.exp(lg), 1e-12)
    idx = np.array([0, 4999, 5050, 9999])
    og = c_oracle.OracleMlp(gw, gb)
    for i in idx:
        xs = engine.draw_samples(xhat[i:i + 1], S, 0.5, 1.0, seed=0, y_index_base=int(i)).astype(np.float64)
        want = c_oracle.is_estimate_given_samples(og, grid[i:i + 1], xhat[i:i + 1], xs, 0.1, 0.5, 1.0, True,
                                                  c_oracle.ORDER_KERNEL, 8)
        np.testing.assert_allclose(lg[i], want[0], rtol=1e-12)


def test_multi_sd_shares_one_generator_pass(engine, make_nets):
    """rt_is_estimate_multi (model_selection's inner loop, main_density_est.py:207-219): every row is
    bit-identical to the single-sd call, in log and linear mode, across chunked and unchunked shapes."""
    gw, gb, hw, hb = make_nets(3, 6)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    sds = [0.05, 0.1, 0.5]
    for n, S in ((5, 2000), (700, 130)):
        y = np.random.RandomState(n).standard_normal((n, 6)) * 0.1
        xhat = engine.mlp_forward(H, y.astype(np.float32))
        for log in (True, False):
            multi = engine.is_estimate_multi(y, xhat, S, sds, 0.1, 1.0, seed=0, log=log)
            assert multi.shape == (3, n)
            for k, sd in enumerate(sds):
                assert np.array_equal(multi[k], engine.is_estimate(y, xhat, S, sd, 0.1, 1.0, seed=0, log=log))
    engine.reset_counters()
    engine.is_estimate_multi(y, xhat, S, sds, 0.1, 1.0)
    c = engine.counters()
    assert c["sample_evals"] == 700 * 130 and c["launches"] == 1     # one G pass, not three
    with pytest.raises(_lib.RtError):
        engine.is_estimate_multi(y, xhat, S, [0.1] * 9, 0.1)
    with pytest.raises(_lib.RtError):
        engine.is_estimate_multi(y, xhat, S, [0.1, -1.0], 0.1)


@pytest.mark.parametrize("dx,dy", DIMS)
def test_jacobian_bit_exact(engine, make_nets, dx, dy):
    """rt_mlp_jacobian (get_jacobian, main_density_est.py:252-263) == the C oracle's forward-mode Jacobian in kernel
    order, bit for bit, for ragged row counts around the 32-row pass; close to the natural-order / numpy oracles."""
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    og, oh = c_oracle.OracleMlp(gw, gb), c_oracle.OracleMlp(hw, hb)
    rng = np.random.RandomState(dx * 7 + dy)
    for n in (1, 31, 32, 33, 70):
        x = rng.standard_normal((n, dx)).astype(np.float32)
        out, J = engine.mlp_jacobian(G, x)
        assert J.shape == (n, dy, dx)
        wout, wJ = og.jacobian(x, c_oracle.ORDER_KERNEL, nthreads=8)
        assert np.array_equal(out, wout)
        assert np.array_equal(J, wJ)
        assert np.array_equal(out, engine.mlp_forward(G, x))
    nat = is_ref.get_jacobian(is_ref.DenseNet(gw, gb), x)
    assert np.abs(J - nat).max() <= 1e-5 * np.abs(nat).max()
    y = rng.standard_normal((40, dy)).astype(np.float32)
    out, J = engine.mlp_jacobian(H, y)
    wout, wJ = oh.jacobian(y, c_oracle.ORDER_KERNEL, nthreads=8)
    assert np.array_equal(out, wout) and np.array_equal(J, wJ)
    assert engine.mlp_jacobian(G, np.zeros((0, dx), np.float32))[1].shape == (0, dy, dx)


def test_jacobian_other_widths_and_limits(engine):
    from conftest import glorot
    for dims in ([4, 128, 128, 3], [2, 100, 200, 7], [3, 256, 256, 256, 2]):
        ws, bs = glorot(dims, 41, 42)
        engine.set_mlp(G, ws, bs)
        x = np.random.RandomState(1).standard_normal((45, dims[0])).astype(np.float32)
        out, J = engine.mlp_jacobian(G, x)
        nat_out, nat_J = c_oracle.OracleMlp(ws, bs).jacobian(x, c_oracle.ORDER_NATURAL, nthreads=8)
        assert np.abs(out - nat_out).max() <= 4e-6 * np.abs(nat_out).max()
        assert np.abs(J - nat_J).max() <= 1e-5 * np.abs(nat_J).max()
    ws, bs = glorot([2, 1024, 1024, 2], 43, 44)
    engine.set_mlp(G, ws, bs)
    with pytest.raises(_lib.RtError):          # 32-row passes (width 1024) cannot pair primal and tangent tiles
        engine.mlp_jacobian(G, np.zeros((3, 2), np.float32))


@pytest.mark.parametrize("name", ["indep_gmm_2_2", "arem_3_6", "ypmsd_20_90", "df3_biased_2_2"])
def test_jacobian_golden(engine, name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    gw, gb, hw, hb = golden_nets(f)
    engine.set_mlp(G, gw, gb)
    out, J = engine.mlp_jacobian(G, f["xhat"])
    np.testing.assert_allclose(out, f["g_of_xhat"], rtol=2e-5, atol=1e-7)
    assert np.abs(J - f["jac_at_xhat"]).max() <= 1e-5 * np.abs(f["jac_at_xhat"]).max()


@pytest.mark.parametrize("S", [1000, 1024, 650, 37, 40000])
def test_pair_ranges_merge_to_the_estimate(engine, make_nets, S):
    """rt_is_pairs over any split of a point's E pairs + rt_is_merge_pairs == rt_is_estimate_multi, bit for bit: what lets
    ranks split ONE query point's samples (second axis).  S with packed leftovers (1000 = 15 x 64 + 40 is
    not packed, 650 and 37 are), a whole number of passes, and the reference's 40000."""
    gw, gb, hw, hb = make_nets(3, 6)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    y = np.random.RandomState(9).standard_normal((3, 6)) * 0.1
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    sds = [0.1, 0.4]
    kw = dict(df=1.0, seed=5, y_index_base=17)
    E = engine.is_pair_count(S)
    assert E == S // 64 + (1 if S % 64 else 0)
    cuts = sorted(set([0, E // 3, E // 3, (2 * E) // 3, E - 1, E]))
    parts = [engine.is_pairs(y, xhat, S, sds, 0.1, lo, hi, **kw) for lo, hi in zip(cuts, cuts[1:])]
    pairs = np.concatenate(parts, axis=1)
    assert pairs.shape == (3, E, 2, 2)
    assert np.array_equal(pairs, engine.is_pairs(y, xhat, S, sds, 0.1, 0, E, **kw))
    assert engine.is_pairs(y, xhat, S, sds, 0.1, 2 if E > 2 else 0, 2 if E > 2 else 0, **kw).shape[1] == 0      # empty range
    for log in (True, False):
        want = engine.is_estimate_multi(y, xhat, S, sds, 0.1, log=log, **kw)
        assert np.array_equal(engine.is_merge_pairs(pairs, S, log=log), want)
    with pytest.raises(_lib.RtError):
        engine.is_pairs(y, xhat, S, sds, 0.1, 0, E + 1, **kw)


def _random_case(rng):
    n_hidden = int(rng.randint(1, 5))
    top = int(rng.choice([128, 256, 512, 512, 1024]))
    widths = [int(rng.randint(max(1, top // 2 - 30), top + 1)) for _ in range(n_hidden)]
    widths[int(rng.randint(n_hidden))] = int(rng.randint(top // 2 + 1, top + 1))      # pins the kernel width class
    dx, dy = int(rng.randint(1, 25)), int(rng.randint(1, 101))
    return dict(dims=[dx] + widths + [dy], N=int(rng.randint(1, 10)), S=int(rng.randint(1, 300)),
                df=float(rng.choice([1.0, 2.0, 3.0, 2.5, 7.3])), sd=float(rng.choice([0.05, 0.3, 1.0])),
                scale=float(rng.choice([0.01, 0.1, 0.5, 1.0])), alpha=float(rng.choice([0.2, 0.2, 0.0, 1.0, 0.37])),
                base=int(rng.choice([0, 11, 2 ** 32 - 3, 2 ** 40 + 5])), seed=int(rng.randint(0, 2 ** 31)),
                log=bool(rng.randint(2)))


@pytest.mark.parametrize("case_seed", range(64))
def test_random_shapes_against_the_oracle(engine, case_seed):
    """Seeded random network shapes (1-4 hidden layers of unequal widths up to 1024, dx up to 24, dy up to 100, any
    leaky-ReLU slope), sample counts, df, scale, sd, seeds and global index bases (beyond 2^32): forward bit-exact, the
    estimate on the GPU's own draws equal to the oracle's scoring of those draws, several sd values in one pass, and
    the result independent of the call the point sits in."""
    rng = np.random.RandomState(1000 + case_seed)
    c = _random_case(rng)
    dims, N, S = c["dims"], c["N"], c["S"]
    gw, gb = is_ref.glorot_mlp(dims, rng)
    gb = [(rng.standard_normal(b.shape) * 0.05).astype(np.float32) for b in gb]
    engine.set_mlp(G, gw, gb, alpha=c["alpha"])
    og = c_oracle.OracleMlp(gw, gb, c["alpha"])
    x = (rng.standard_normal((70, dims[0])) * 2).astype(np.float32)
    assert np.array_equal(engine.mlp_forward(G, x), og.forward(x, order=c_oracle.ORDER_KERNEL)), c
    if max(dims[1:-1]) <= 512:                                   # the Jacobian kernel runs 64-column passes only
        out, J = engine.mlp_jacobian(G, x[:37])
        wout, wJ = og.jacobian(x[:37], c_oracle.ORDER_KERNEL)
        assert np.array_equal(out, wout) and np.array_equal(J, wJ), c
    y = rng.standard_normal((N, dims[-1])) * 0.3
    xhat = (rng.standard_normal((N, dims[0])) * 0.3).astype(np.float32)
    kw = dict(df=c["df"], seed=c["seed"], y_index_base=c["base"])
    got = engine.is_estimate(y, xhat, S, c["sd"], c["scale"], log=c["log"], **kw)
    xs = engine.draw_samples(xhat, S, c["scale"], **kw).astype(np.float64)
    want = c_oracle.is_estimate_given_samples(og, y, xhat, xs, c["sd"], c["scale"], c["df"], c["log"], c_oracle.ORDER_KERNEL)
    if c["log"]:
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13, err_msg=str(c))
    else:
        # log=False: the reference (and the oracle, which follows it) multiplies p(y|x_i) and w_i in linear space, where
        # terms below ~1e-308 flush to zero one by one; the GPU reduces in log space and exponentiates once, so down in
        # the underflow zone it can return a denormal-scale value where the reference says exactly 0
        ok = want > 1e-280
        assert np.all(got[~ok] < 1e-270), c
        np.testing.assert_allclose(np.log(got[ok]), np.log(want[ok]), rtol=1e-11, atol=1e-9, err_msg=str(c))
    multi = engine.is_estimate_multi(y, xhat, S, [c["sd"], 2 * c["sd"]], c["scale"], log=c["log"], **kw)
    assert np.array_equal(multi[0], got), c
    k = N // 2
    alone = engine.is_estimate(y[k:k + 1], xhat[k:k + 1], S, c["sd"], c["scale"], log=c["log"], df=c["df"],
                               seed=c["seed"], y_index_base=c["base"] + k)
    assert np.array_equal(alone, got[k:k + 1]), c


# ---------------------------------------------------------------------------------------------------------------
# Round 2
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dx,dy", [(2, 2), (20, 90)])
@pytest.mark.parametrize("S", [7, 16, 32])
def test_first_pass_is_a_leftover_pass(engine, make_nets, dx, dy, S):
    """S <= 32: every pass of the launch is a packed leftover pass (p_main == 0), so a workgroup's first write to the x0
    image follows its one-time zero-fill with nothing but the barrier after that fill in between; more query points
    than CUs so that every workgroup runs.  Same for a pair range that holds only the last (leftover) entry."""
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    og = c_oracle.OracleMlp(gw, gb)
    rng = np.random.RandomState(100 + S)
    N = 3 * engine.cu_count() + 5
    y = rng.standard_normal((N, dy)) * 0.3
    xhat = (rng.standard_normal((N, dx)) * 0.3).astype(np.float32)
    got = engine.is_estimate(y, xhat, S, 0.5, 0.1, seed=3)
    xs = engine.draw_samples(xhat, S, 0.1, seed=3).astype(np.float64)
    want = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.5, 0.1, 1.0, True, c_oracle.ORDER_KERNEL, nthreads=8)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    # given samples take the all-threads write path of the same phase
    np.testing.assert_allclose(engine.is_estimate_given_samples(y, xhat, xs, 0.5, 0.1), want, rtol=1e-12, atol=1e-13)
    # the last entry alone, at a sample count with full passes before it
    S2 = 64 + S
    E = engine.is_pair_count(S2)
    full = engine.is_pairs(y, xhat, S2, [0.5], 0.1, 0, E, seed=3)
    last = engine.is_pairs(y, xhat, S2, [0.5], 0.1, E - 1, E, seed=3)
    assert np.array_equal(last[:, 0], full[:, E - 1])


def _nan_generator(dx, dy):
    """A generator that is finite near the origin and overflows far from it: two hidden layers of gain 1e19 each on
    x_0 and on -x_0 (through the leaky-ReLU: 1e38 x_0 on the positive branch, -4e36 x_0 on the other), summed by the
    output layer: finite for small x_0, +inf from x_0 ~ 3.4 on, inf + (-inf) = NaN from x_0 ~ 1e2 on."""
    w0 = np.zeros((dx, 128), np.float32); w0[0, 0] = 1e19; w0[0, 1] = -1e19
    w1 = np.zeros((128, 128), np.float32); w1[0, 0] = 1e19; w1[1, 1] = 1e19
    w2 = np.zeros((128, dy), np.float32); w2[0, :] = 1.0; w2[1, :] = 1.0
    return [w0, w1, w2], [np.zeros(128, np.float32), np.zeros(128, np.float32), np.zeros(dy, np.float32)]


def test_non_finite_scores_follow_the_reference(engine):
    """-inf scores (G overflowed to +-inf) weigh 0; a NaN score makes the point's estimate NaN, as the reference's
    exp(v - max) / mean do (main_density_est.py:342,345); every score -inf gives NaN under log=True ((-inf) - (-inf))
    and 0 under log=False.  The C oracle follows the reference literally, so it is the checker."""
    ws, bs = _nan_generator(2, 2)
    engine.set_mlp(G, ws, bs)
    og = c_oracle.OracleMlp(ws, bs)
    y = np.array([[0.1, -0.2], [0.0, 0.3], [0.2, 0.2], [1.0, 1.0]])
    xhat = np.zeros((4, 2), np.float32)
    S = 100
    xs = np.zeros((4, S, 2))
    xs[:, :, 0] = 1e-38                               # G = 0.96: finite scores
    xs[:, :, 1] = np.linspace(-1, 1, S)
    xs[1, 5, 0] = 10.0                                # the positive branch alone overflows: G = +inf, score -inf
    xs[2, 70, 0] = 100.0                              # both branches overflow: G = inf + (-inf) = NaN
    xs[3, :, 0] = 10.0                                # every sample -inf
    out = og.forward(xs.reshape(-1, 2).astype(np.float32), order=c_oracle.ORDER_KERNEL).reshape(4, S, 2)
    assert np.isfinite(out[0]).all() and np.isinf(out[1, 5]).all() and np.isnan(out[2, 70]).all()
    for log in (True, False):
        with np.errstate(all="ignore"):
            want = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.5, 0.5, 1.0, log, c_oracle.ORDER_KERNEL)
        got = engine.is_estimate_given_samples(y, xhat, xs, 0.5, 0.5, 1.0, log=log)
        assert np.isfinite(got[:2]).all() and np.isnan(got[2]) and np.isnan(want[2]), (got, want)
        np.testing.assert_allclose(got[:2], want[:2], rtol=1e-12)
        if log:
            assert np.isnan(got[3]) and np.isnan(want[3])
        else:
            assert got[3] == 0.0 and want[3] == 0.0
    # the same through the packed leftover passes (S <= 32) and through the merge of pairs
    got = engine.is_estimate_given_samples(y, xhat, np.ascontiguousarray(xs[:, 60:80]), 0.5, 0.5, 1.0)
    assert np.isfinite(got[:2]).all() and np.isnan(got[2]) and np.isnan(got[3])


@pytest.mark.parametrize("dx,dy,S,sd,scale", [(20, 90, 160000, 0.5, 0.1), (20, 90, 640000, 0.5, 0.1),
                                              (8, 21, 40000, 0.1, 0.1), (5, 9, 40000, 0.1, 0.1)])
def test_large_sample_counts_of_the_baseline_configs(engine, make_nets, dx, dy, S, sd, scale):
    """BASELINE configs 3-5 at the sample counts that define them (YearPredictionMSD 20/90 at S = 160 000 / 640 000,
    HEPMASS 8/21 and CASP 5/9 at 40 000): the sampler instantiation that shares a sample's Philox calls over four
    threads sees sample indices up to 6.4e5 here.  Size-independent properties on all points — finite, exp(log) equals
    the linear answer, a point evaluated alone is bit-identical, pair ranges merge to the estimate — and the oracle (kernel
    order, 1e-12) on the GPU's own draws for one or two points."""
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    rng = np.random.RandomState(S % 1000 + dx)
    N = 32 if S <= 160000 else 16
    y = rng.standard_normal((N, dy)) * 0.3
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    base = 262144 - 5
    lg = engine.is_estimate(y, xhat, S, sd, scale, y_index_base=base)
    lin = engine.is_estimate(y, xhat, S, sd, scale, log=False, y_index_base=base)
    assert np.all(np.isfinite(lg))
    ok = lg > -600
    np.testing.assert_allclose(np.log(lin[ok]), lg[ok], rtol=1e-12)
    k = N - 3
    alone = engine.is_estimate(y[k:k + 1], xhat[k:k + 1], S, sd, scale, y_index_base=base + k)
    assert np.array_equal(alone, lg[k:k + 1])
    # pair ranges of a few points, split at uneven places, merge to the same bits
    E = engine.is_pair_count(S)
    cuts = [0, E // 3, E // 3 + 1, E - 1, E]
    sub = slice(0, 4)
    parts = [engine.is_pairs(y[sub], xhat[sub], S, [sd], scale, lo, hi, y_index_base=base)
             for lo, hi in zip(cuts[:-1], cuts[1:])]
    merged = engine.is_merge_pairs(np.concatenate(parts, axis=1), S)[0]
    assert np.array_equal(merged, lg[sub])
    # oracle on the same draws
    n_or = 2 if S <= 160000 else 1
    og = c_oracle.OracleMlp(gw, gb)
    xs = engine.draw_samples(xhat[:n_or], S, scale, y_index_base=base).astype(np.float64)
    want = c_oracle.is_estimate_given_samples(og, y[:n_or], xhat[:n_or], xs, sd, scale, 1.0, True,
                                              c_oracle.ORDER_KERNEL, nthreads=16)
    np.testing.assert_allclose(lg[:n_or], want, rtol=1e-12, atol=1e-13)
    # the oracle's own restatement of the sampler at the largest sample indices
    tail = c_oracle.draw_samples(xhat[:1], base, S, scale, 1.0, 0)[0, -4096:]
    assert np.mean(np.abs(tail - xs[0, -4096:]) <= 2e-3 * np.maximum(1.0, np.abs(tail))) >= 0.9999


PHILOX_KAT = [   # Random123 kat_vectors, philox4x32 with 10 rounds: counter[4], key[2] -> output[4]
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers_on_the_gpu(engine):
    """The sampler's generator IS the published Philox4x32-10: the three Random123 known-answer vectors, and 10^5 random
    (counter, key) rows against the C restatement (itself pinned to the same vectors in tests/test_oracle.py)."""
    rows = np.array([c + k for c, k, _ in PHILOX_KAT], dtype=np.uint32)
    got = engine.philox4x32_10(rows)
    assert [list(map(int, r)) for r in got] == [o for _, _, o in PHILOX_KAT]
    rnd = np.random.RandomState(9).randint(0, 2 ** 32, size=(100000, 6), dtype=np.uint64).astype(np.uint32)
    got = engine.philox4x32_10(rnd)
    want = np.stack([c_oracle.philox4x32_10(r[:4], r[4:]) for r in rnd[:2000]])
    assert np.array_equal(got[:2000], want)
    # the draws are that generator at counter = (sample, call, y lo, y hi), key = seed: first coordinate of sample i of
    # query point n, decoded by hand from the raw words (df = 1, dx = 2: normals 0 = chi, 1 and 2 = coordinates)
    engine.set_mlp(G, *[[np.zeros((2, 8), np.float32), np.zeros((8, 2), np.float32)],
                        [np.zeros(8, np.float32), np.zeros(2, np.float32)]])
    base, seed = (7 << 32) + 5, (3 << 32) + 11
    xs = engine.draw_samples(np.zeros((2, 2), np.float32), 64, 1.0, 1.0, seed=seed, y_index_base=base)
    ck = np.array([[i, 0, (base + n) & 0xffffffff, (base + n) >> 32, seed & 0xffffffff, seed >> 32]
                   for n in range(2) for i in range(64)], dtype=np.uint32)
    w = engine.philox4x32_10(ck).astype(np.float64)
    u = ((w // 512) + 0.5) / 8388608.0
    r0 = np.sqrt(-2 * np.log(u[:, 0])); g0, g1 = r0 * np.cos(2 * np.pi * u[:, 1]), r0 * np.sin(2 * np.pi * u[:, 1])
    g2 = np.sqrt(-2 * np.log(u[:, 2])) * np.cos(2 * np.pi * u[:, 3])
    np.testing.assert_allclose(xs.reshape(-1, 2)[:, 0], g1 / np.abs(g0), rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(xs.reshape(-1, 2)[:, 1], g2 / np.abs(g0), rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("df", [2.5, 7.3, 0.7, 40.0])
def test_gamma_path_draws_match_the_restatement(engine, make_nets, df):
    """Non-integer df (and integer df > 32): chi2 = 2 Gamma(df/2) by Marsaglia-Tsang — compared value by value with
    the C restatement (oracle/is_oracle.c chi2_over_df_gamma), like the integer-df draws, not only by KS."""
    gw, gb, _, _ = make_nets(5, 9)
    engine.set_mlp(G, gw, gb)
    xhat = (np.random.RandomState(2).standard_normal((3, 5)) * 0.3).astype(np.float32)
    S = 20000
    xs = engine.draw_samples(xhat, S, 0.5, df, seed=21, y_index_base=1 << 33)
    xo = c_oracle.draw_samples(xhat, 1 << 33, S, 0.5, df, 21)
    close = np.isclose(xs, xo, rtol=2e-3, atol=1e-5).all(axis=2)
    assert close.mean() > 0.999, (df, close.mean())
    u = (xs[0].astype(np.float64) - xhat[0]) / 0.5
    assert stats.kstest(u[:, 3], stats.t(df).cdf).pvalue > 1e-3


def test_xcd_balance_changes_no_result(engine, make_nets, monkeypatch):
    """The XCDs of one chip run at slightly different clocks; after every long launch the library re-derives, on the
    device, the share of the passes each of the eight workgroup classes (blockIdx % 8) gets next.  Only WHICH workgroup
    runs a pass changes: a context with the balancing off (RT_IS_BALANCE=0) gives the same bits, launch after launch,
    and the shares stay a partition of unity within their clamps."""
    gw, gb, hw, hb = make_nets(2, 2)
    rng = np.random.RandomState(77)
    N, S = 4096, 6400 * 3            # 1200 passes per workgroup: long enough to update the shares
    y = rng.standard_normal((N, 2)) * 0.5
    monkeypatch.setenv("RT_IS_BALANCE", "0")
    plain = _lib.Engine(0)
    monkeypatch.delenv("RT_IS_BALANCE")
    try:
        plain.set_mlp(G, gw, gb)
        plain.set_mlp(H, hw, hb)
        engine.set_mlp(G, gw, gb)
        engine.set_mlp(H, hw, hb)
        xhat = engine.mlp_forward(H, y.astype(np.float32))
        want = plain.is_estimate(y, xhat, S, 0.1, 0.5)
        assert np.allclose(plain.xcd_shares(), 0.125, rtol=0, atol=0)          # off: equal shares, never updated
        for _ in range(3):
            got = engine.is_estimate(y, xhat, S, 0.1, 0.5)
            assert np.array_equal(got, want)
            sh = engine.xcd_shares()
            assert abs(sh.sum() - 1.0) < 1e-12 and sh.min() >= 0.8 / 8 - 1e-12 and sh.max() <= 1.25 / 8 + 1e-12
        # a short launch uses the shares without touching them, and packed leftover passes are dealt by them too
        before = engine.xcd_shares()
        a = engine.is_estimate(y[:300], xhat[:300], 1000 + 16, 0.1, 0.5, log=False)
        assert np.array_equal(a, plain.is_estimate(y[:300], xhat[:300], 1000 + 16, 0.1, 0.5, log=False))
        assert np.array_equal(engine.xcd_shares(), before)
    finally:
        plain.close()


@pytest.mark.parametrize("hid", [64, 100, 128, 256])
@pytest.mark.parametrize("dx,dy", [(2, 2), (8, 21), (5, 32)])
def test_one_and_two_wave_workgroups_equal_the_four_wave_kernel(engine, make_nets, monkeypatch, hid, dx, dy):
    """Hidden widths <= 128 (256) with an output dim <= 32 run as one-wave (two-wave) workgroups, four (two) to a CU.
    Every entry point must give the SAME BITS as the four-wave kernel (a context created under RT_IS_NARROW=0, which
    zero-pads width 64 / 100 to 128) — forward, Jacobian, the estimate on Philox draws with packed leftover passes,
    on given samples, for several sd values, as pair ranges — and agree with the oracle in kernel order."""
    gw, gb = glorot_nets(dx, hid, dy, 7)
    monkeypatch.setenv("RT_IS_NARROW", "0")
    wide = _lib.Engine(0)
    try:
        wide.set_mlp(G, gw, gb)
        monkeypatch.delenv("RT_IS_NARROW")
        engine.set_mlp(G, gw, gb)
        og = c_oracle.OracleMlp(gw, gb)
        rng = np.random.RandomState(hid + dx)
        x = (rng.standard_normal((150, dx)) * 2).astype(np.float32)
        f = engine.mlp_forward(G, x)
        assert np.array_equal(f, wide.mlp_forward(G, x))
        assert np.array_equal(f, og.forward(x, order=c_oracle.ORDER_KERNEL))
        o1, j1 = engine.mlp_jacobian(G, x[:70])
        o2, j2 = wide.mlp_jacobian(G, x[:70])
        assert np.array_equal(o1, o2) and np.array_equal(j1, j2)
        N = 4 * engine.cu_count() + 37                      # more passes than workgroups of either kind
        y = rng.standard_normal((N, dy)) * 0.3
        xhat = (rng.standard_normal((N, dx)) * 0.3).astype(np.float32)
        for S, log in ((64 * 3 + 16, True), (40, False), (200, True)):
            kw = dict(df=1.0, seed=9, y_index_base=5)
            got = engine.is_estimate(y, xhat, S, 0.3, 0.2, log=log, **kw)
            assert np.array_equal(got, wide.is_estimate(y, xhat, S, 0.3, 0.2, log=log, **kw)), (S, log)
        S = 64 * 3 + 16
        got = engine.is_estimate(y, xhat, S, 0.3, 0.2, **kw)
        xs = engine.draw_samples(xhat[:40], S, 0.2, seed=9, y_index_base=5).astype(np.float64)
        assert np.array_equal(xs, wide.draw_samples(xhat[:40], S, 0.2, seed=9, y_index_base=5).astype(np.float64))
        want = c_oracle.is_estimate_given_samples(og, y[:40], xhat[:40], xs, 0.3, 0.2, 1.0, True, c_oracle.ORDER_KERNEL, 8)
        np.testing.assert_allclose(got[:40], want, rtol=1e-12, atol=1e-13)
        gs = engine.is_estimate_given_samples(y[:40], xhat[:40], xs, 0.3, 0.2)
        assert np.array_equal(gs, wide.is_estimate_given_samples(y[:40], xhat[:40], xs, 0.3, 0.2))
        np.testing.assert_allclose(gs, want, rtol=1e-12, atol=1e-13)
        multi = engine.is_estimate_multi(y, xhat, S, [0.3, 0.1, 0.7], 0.2, seed=9, y_index_base=5)
        assert np.array_equal(multi, wide.is_estimate_multi(y, xhat, S, [0.3, 0.1, 0.7], 0.2, seed=9, y_index_base=5))
        assert np.array_equal(multi[0], got)
        E = engine.is_pair_count(S)
        assert E == wide.is_pair_count(S) == 4
        parts = [engine.is_pairs(y[:50], xhat[:50], S, [0.3], 0.2, lo, hi, seed=9, y_index_base=5)
                 for lo, hi in ((0, 1), (1, 3), (3, 4))]
        whole = wide.is_pairs(y[:50], xhat[:50], S, [0.3], 0.2, 0, E, seed=9, y_index_base=5)
        assert np.array_equal(np.concatenate(parts, axis=1), whole)
        assert np.array_equal(engine.is_merge_pairs(whole, S)[0], got[:50])
        # a many-dimensional non-integer-df draw goes through the one-thread-per-sample sampler here
        a = engine.is_estimate(y[:64], xhat[:64], 300, 0.3, 0.2, df=2.5, seed=3)
        assert np.array_equal(a, wide.is_estimate(y[:64], xhat[:64], 300, 0.3, 0.2, df=2.5, seed=3))
    finally:
        wide.close()


def test_two_contexts_on_two_threads(make_nets):
    """rt_is.h: 'a context is not re-entrant; different contexts may be used from different threads'.  Two contexts on
    HIP device 0, each driven by its own thread at the same time (ctypes releases the GIL, so the calls do overlap on the
    GPU), with different generators: every result equals what the same context returns when it runs alone."""
    import threading
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    cases = []
    for dx, dy, S in ((2, 2, 5000), (8, 21, 3000)):
        gw, gb, hw, hb = make_nets(dx, dy)
        eng = _lib.Engine(0)
        eng.set_mlp(G, gw, gb)
        eng.set_mlp(H, hw, hb)
        y = np.random.RandomState(dx).standard_normal((300, dy))
        xhat = eng.mlp_forward(H, y.astype(np.float32))
        alone = (eng.is_estimate(y, xhat, S, 0.1, 0.5), eng.mlp_jacobian(G, xhat)[1],
                 eng.is_estimate_multi(y, xhat, S, [0.05, 0.2], 0.5))
        cases.append((eng, y, xhat, S, alone))
    got, errs = {}, []

    def work(k):
        eng, y, xhat, S, _ = cases[k]
        try:
            for _ in range(3):
                got[k] = (eng.is_estimate(y, xhat, S, 0.1, 0.5), eng.mlp_jacobian(G, xhat)[1],
                          eng.is_estimate_multi(y, xhat, S, [0.05, 0.2], 0.5))
        except Exception as e:          # surfaced below: an exception in a thread does not fail a test by itself
            errs.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(240)
    assert not errs, errs
    assert not any(t.is_alive() for t in threads)
    for k, (eng, _, _, _, alone) in enumerate(cases):
        for a, b in zip(alone, got[k]):
            assert np.array_equal(a, b)
        eng.close()


def test_first_long_call_of_a_process_measures_the_shares_on_a_first_block(tmp_path):
    """A process's first long call on a GPU has no measured XCD shares yet: it runs the minimum that updates them
    (320 passes per workgroup) as a block of its own, and the rest of the call on the measured shares — two launches, the
    next call one; the bits are those of a context with the balancing off.  (Fresh process: contexts of one process hand
    their shares on.)"""
    import subprocess, sys, textwrap
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
        N, S = 2048, 19200                                   # 300 passes per point; 320 x 256 passes = 274 points first
        y = np.random.RandomState(3).standard_normal((N, 2)) * 0.5
        xhat = (y * 0.3).astype(np.float32)
        eng = _lib.Engine(0)
        eng.set_mlp(0, gw, gb)
        assert np.array_equal(eng.xcd_shares(), np.full(8, 0.125))
        eng.reset_counters()
        first = eng.is_estimate(y, xhat, S, 0.1, 0.5)
        c1 = eng.counters()
        sh = eng.xcd_shares()
        second = eng.is_estimate(y, xhat, S, 0.1, 0.5)
        c2 = eng.counters()
        assert c1["launches"] == 2 and c2["launches"] == 3, (c1, c2)
        assert c1["sample_evals"] == N * S and c2["sample_evals"] == 2 * N * S
        assert abs(sh.sum() - 1.0) < 1e-12 and not np.array_equal(sh, np.full(8, 0.125))
        short = eng.is_estimate(y[:500], xhat[:500], S, 0.1, 0.5)      # shares known: one launch
        assert eng.counters()["launches"] == 4
        os.environ["RT_IS_BALANCE"] = "0"
        plain = _lib.Engine(0)
        plain.set_mlp(0, gw, gb)
        want = plain.is_estimate(y, xhat, S, 0.1, 0.5)
        assert plain.counters()["launches"] == 1
        assert np.array_equal(first, want) and np.array_equal(second, want) and np.array_equal(short, want[:500])
        # RT_IS_FIRST_BLOCK=0 (bench.py sets it: one launch per step): the first long call stays one launch and the shares
        # are measured for the next call only
        del os.environ["RT_IS_BALANCE"]
        os.environ["RT_IS_FIRST_BLOCK"] = "0"
        whole = _lib.Engine(0)                               # (eng is still alive: nothing has been handed on yet)
        whole.set_mlp(0, gw, gb)
        assert np.array_equal(whole.is_estimate(y, xhat, S, 0.1, 0.5), want) and whole.counters()["launches"] == 1
        assert not np.array_equal(whole.xcd_shares(), np.full(8, 0.125))
        print("calibration ok")
    """))
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=280, cwd=str(tmp_path),
                         env={k: v for k, v in os.environ.items() if k not in ("RT_IS_BALANCE", "RT_IS_FIRST_BLOCK")})
    assert res.returncode == 0 and "calibration ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.parametrize("dims", [
    [104, 512, 512, 128],      # width 512: the x0 image of 104 input dims fills the 160 KB of LDS to within 96 bytes
    [128, 256, 256, 64],       # the ABI's largest input dim (MAX_IN = 128) at width 256, with an output dim whose four
                               # partial-sum planes are as large as the activation image
    [104, 256, 256, 128],      # the largest output dim (MAX_OUT = 128: 128 KB of partial sums) leaves room for 104 input dims
    [128, 128, 128, 32],       # ... as one-wave workgroups (output dim <= 32)
    [128, 64, 64, 1],          # ... at width 64
    [1, 1024, 1024, 128],      # 32-sample passes, one input dim, the largest output dim
])
def test_largest_dims_the_kernels_take(engine, dims):
    """The edges of the supported shapes: forward and Jacobian bit-exact, the estimate on the GPU's own draws equal to
    the oracle's scoring of them — packed leftover passes included (S = 200 = 3 x 64 + 8)."""
    rng = np.random.RandomState(sum(dims))
    gw, gb = is_ref.glorot_mlp(dims, rng)
    gb = [(rng.standard_normal(b.shape) * 0.05).astype(np.float32) for b in gb]
    engine.set_mlp(G, gw, gb)
    og = c_oracle.OracleMlp(gw, gb, 0.2)
    x = (rng.standard_normal((70, dims[0])) * 0.5).astype(np.float32)
    assert np.array_equal(engine.mlp_forward(G, x), og.forward(x, order=c_oracle.ORDER_KERNEL))
    if max(dims[1:-1]) <= 512:
        out, J = engine.mlp_jacobian(G, x[:33])
        wout, wJ = og.jacobian(x[:33], c_oracle.ORDER_KERNEL)
        assert np.array_equal(out, wout) and np.array_equal(J, wJ)
    N, S = 5, 200
    y = rng.standard_normal((N, dims[-1])) * 0.3
    xhat = (rng.standard_normal((N, dims[0])) * 0.3).astype(np.float32)
    for df in (1.0, 2.5):
        got = engine.is_estimate(y, xhat, S, 0.5, 0.1, df=df, seed=9)
        xs = engine.draw_samples(xhat, S, 0.1, df=df, seed=9).astype(np.float64)
        want = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.5, 0.1, df, True, c_oracle.ORDER_KERNEL)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(engine.is_estimate_given_samples(y, xhat, xs, 0.5, 0.1, df), want, rtol=1e-12, atol=1e-13)


def test_shapes_beyond_the_lds_budget_are_refused(engine):
    """One input dim more than fits (k0 = 112 at width 512: 164 KB of LDS), dims above 128 and widths above 1024 are
    refused by rt_set_mlp with RT_EUNSUPPORTED — nothing is launched; the context keeps working."""
    z = lambda *s: np.zeros(s, np.float32)
    fresh = _lib.Engine(0)
    try:
        for dims in ([105, 512, 128], [128, 256, 128], [129, 128, 2], [2, 128, 129], [2, 1025, 2]):
            with pytest.raises(_lib.RtError) as e:
                fresh.set_mlp(G, [z(a, b) for a, b in zip(dims[:-1], dims[1:])], [z(b) for b in dims[1:]])
            assert e.value.code == 4, (dims, str(e.value))
        fresh.set_mlp(G, [z(2, 128), z(128, 2)], [z(128), z(2)])
        assert np.array_equal(fresh.mlp_forward(G, z(3, 2)), z(3, 2))
    finally:
        fresh.close()


def test_scorer_log1p_against_libm(engine):
    """The scorer's own log(1 + a) (rt_kernels.hpp log1p_pos, ~45 float64 instructions instead of the device library's
    ~105) against numpy's log1p — libm, < 1 ulp — on the arguments the Student-t proposal produces and far beyond:
    <= 3 ulp of the result everywhere (measured: 2 against long double on the host, scripts/log1p_check.c), exact at 0,
    below 2^-53 and at +inf, NaN carried."""
    rng = np.random.RandomState(11)
    a = np.concatenate([
        np.exp(rng.uniform(-700, 700, 400000)), rng.uniform(0, 3, 400000), np.exp(rng.uniform(-40, 40, 400000)),
        np.ldexp(1.0 + rng.uniform(-1e-6, 1e-6, 200000), rng.randint(-100, 100, 200000)),
        [0.0, 5e-324, 1e-310, 2.2250738585072014e-308, 1e-17, 2.0 ** -53, 2.0 ** -52, np.sqrt(2.0) - 1.0, 1.0, 1.0 - 2.0 ** -53,
         3.0, 1e154, 1e300, np.finfo(np.float64).max]])
    got = engine.log1p_pos(a)
    want = np.log1p(a)
    ulp = np.spacing(np.abs(want))
    err = np.abs(got - want) / ulp
    assert err.max() <= 3.0, (err.max(), a[err.argmax()])
    tiny = a < 2.0 ** -54
    assert np.array_equal(got[tiny], a[tiny])
    sp = engine.log1p_pos(np.array([np.inf, np.nan, 0.0]))
    assert sp[0] == np.inf and np.isnan(sp[1]) and sp[2] == 0.0
    # odd and even positions run the two forms the kernels use (behind a call / inlined): same bits
    b = np.repeat(a[:100000], 2)
    g2 = engine.log1p_pos(b)
    assert np.array_equal(g2[0::2], g2[1::2])
