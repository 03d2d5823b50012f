"""CPU tests of the oracles themselves (no GPU): the C restatement against the numpy/scipy restatement,
against the committed golden vectors, and against closed forms.  The reference has no tests for this
path; these are the known answers the oracle is pinned by."""
import os

import numpy as np
import pytest
from scipy import stats

from oracle import c_oracle, is_ref

from conftest import glorot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_NAMES = ["indep_gmm_2_2", "arem_3_6", "ypmsd_20_90", "df3_biased_2_2"]


def golden_nets(f):
    """Rebuilds the G / H weights of a golden fixture from its recorded seeds (see oracle/make_golden.py)."""
    dx, dy = int(f["dx"]), int(f["dy"])
    rng = np.random.RandomState(int(f["weight_seed"]))
    gw, gb = is_ref.glorot_mlp([dx] + [512] * 10 + [dy], rng)
    hw, hb = is_ref.glorot_mlp([dy] + [256] * 10 + [dx], rng)
    if int(f["bias_seed"]) >= 0:
        r = np.random.RandomState(int(f["bias_seed"]))
        gb = [r.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in gb]
        hb = [r.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in hb]
    return gw, gb, hw, hb


def golden_samples(f):
    """Re-draws the fixture's proposal samples exactly as the reference does (np.random.seed(0), then per
    query point chisquare + multivariate_normal: main_density_est.py:267, 316-318)."""
    np.random.seed(0)
    xhat = f["xhat"].astype(np.float64)
    xs = np.stack([is_ref.sample_from_qx(xhat[n], int(f["dx"]), float(f["scale"]), float(f["df"]), int(f["S"]))
                   for n in range(xhat.shape[0])])
    assert np.array_equal(xs[:, :8, :], f["xs_head"])   # legacy RandomState stream is frozen (NEP 19)
    return xs


def assert_linear_close(got, want, rtol_log):
    """Linear-space densities span hundreds of orders of magnitude; a relative error in p IS an absolute
    error in log p, so compare logs with the log-space tolerance.  Exact zeros (underflow) must agree."""
    got, want = np.asarray(got), np.asarray(want)
    assert np.array_equal(got == 0.0, want == 0.0)
    nz = want != 0.0
    np.testing.assert_allclose(np.log(got[nz]), np.log(want[nz]), rtol=rtol_log, atol=1e-9)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_c_oracle_matches_golden(name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    gw, gb, hw, hb = golden_nets(f)
    og, oh = c_oracle.OracleMlp(gw, gb), c_oracle.OracleMlp(hw, hb)
    # H forward reproduces the stored proposal centres
    xhat = oh.forward(f["y"].astype(np.float32))
    np.testing.assert_allclose(xhat, f["xhat"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(og.forward(f["xhat"]), f["g_of_xhat"], rtol=2e-5, atol=1e-7)
    xs = golden_samples(f)
    for order in (c_oracle.ORDER_NATURAL, c_oracle.ORDER_KERNEL):
        got = c_oracle.is_estimate_given_samples(og, f["y"], f["xhat"], xs, float(f["sd"]), float(f["scale"]),
                                                 float(f["df"]), True, order, nthreads=8)
        np.testing.assert_allclose(got, f["log_est"], rtol=1e-6)
    lin = c_oracle.is_estimate_given_samples(og, f["y"], f["xhat"], xs, float(f["sd"]), float(f["scale"]),
                                             float(f["df"]), False, c_oracle.ORDER_NATURAL, nthreads=8)
    assert_linear_close(lin, f["lin_est"], 1e-6)


def test_grid_fixture_is_from_reference_util():
    f = np.load(os.path.join(GOLDEN, "indep_gmm_2_2.npz"))
    assert bool(f["grid_from_reference_util"])
    grid = is_ref.create_2d_grid_data(-1.5, 1.5, -1.5, 1.5, n=100)[2]
    assert np.array_equal(grid[::1237][:8], f["y"])


@pytest.mark.parametrize("df", [1.0, 2.0, 3.0, 2.5, 30.0])
def test_t_logpdf_closed_form(df):
    """scipy.stats.t.logpdf(x, df, loc=0, scale) — the reference's call at main_density_est.py:301."""
    for scale in (0.005, 0.1, 0.5, 1.0):
        xs = np.concatenate([np.linspace(-5, 5, 41), [1e-9, -1e3, 1e6 * scale, -1e12]])
        want = stats.t.logpdf(xs, df, loc=0, scale=scale)
        got = np.array([c_oracle.t_logpdf(x, df, scale) for x in xs])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    if df == 1.0:   # Cauchy: -ln(pi) - ln(scale) - ln(1+u^2)
        u = 3.7
        assert abs(c_oracle.t_logpdf(u * 0.5, 1.0, 0.5) - (-np.log(np.pi) - np.log(0.5) - np.log1p(u * u))) < 1e-13


def test_mlp_orders_and_numpy_agree():
    ws, bs = glorot([5, 128, 128, 128, 9], 1, 2)
    x = np.random.RandomState(0).standard_normal((77, 5)).astype(np.float32) * 3
    o = c_oracle.OracleMlp(ws, bs)
    a = o.forward(x, c_oracle.ORDER_NATURAL)
    b = o.forward(x, c_oracle.ORDER_KERNEL, nthreads=4)
    c = is_ref.DenseNet(ws, bs)(x)
    scale = np.abs(c).max()
    assert np.abs(a - c).max() < 1e-5 * scale
    assert np.abs(a - b).max() < 1e-5 * scale
    # threads do not change results
    assert np.array_equal(b, o.forward(x, c_oracle.ORDER_KERNEL, nthreads=1))


def test_leaky_relu_matches_reference_formula():
    # model.py:5-6: max(min(0, a*x), x)
    x = np.array([-3.0, -0.0, 0.0, 1e-30, 2.5, -1e30], dtype=np.float32)
    want = np.maximum(np.minimum(0.0, 0.2 * x), x)
    ws = [np.eye(6, dtype=np.float32), np.eye(6, dtype=np.float32)]
    bs = [np.zeros(6, np.float32), np.zeros(6, np.float32)]
    got = c_oracle.OracleMlp(ws, bs).forward(x[None, :])[0]
    np.testing.assert_array_equal(got, want.astype(np.float32))


def linear_g(dx, dy, hidden, n_hidden, A, b):
    """Weights that make the leaky-ReLU MLP EXACTLY linear, G(x) = x @ A + b: lrelu(t) - lrelu(-t) = 1.2 t,
    so a layer pair [I, -I] ... [[M], [-M]] / 1.2 carries a linear map through any depth."""
    assert hidden >= 2 * max(dx, dy)
    ws, bs = [], []
    w0 = np.zeros((dx, hidden), np.float32)
    w0[:, :dx] = np.eye(dx)
    w0[:, dx:2 * dx] = -np.eye(dx)
    ws.append(w0)
    bs.append(np.zeros(hidden, np.float32))
    for _ in range(n_hidden - 1):
        w = np.zeros((hidden, hidden), np.float32)
        w[:dx, :dx] = np.eye(dx) / 1.2
        w[dx:2 * dx, :dx] = -np.eye(dx) / 1.2
        w[:dx, dx:2 * dx] = -np.eye(dx) / 1.2
        w[dx:2 * dx, dx:2 * dx] = np.eye(dx) / 1.2
        ws.append(w)
        bs.append(np.zeros(hidden, np.float32))
    wl = np.zeros((hidden, dy), np.float32)
    wl[:dx, :] = A / 1.2
    wl[dx:2 * dx, :] = -A / 1.2
    ws.append(wl)
    bs.append(np.asarray(b, np.float32))
    return ws, bs


def gaussian_logpdf(y, mean, cov):
    d = y - mean
    _, logdet = np.linalg.slogdet(cov)
    return -0.5 * (len(y) * np.log(2 * np.pi) + logdet + d @ np.linalg.solve(cov, d))


def test_analytic_linear_generator_known_answer():
    """With G(x) = x A + b and x ~ N(0, I), p(y) = N(y; b, A^T A + sd^2 I) in closed form; the IS estimate
    must converge to it (tests the weights, the Student-t weights and the reduction end to end).
    dx = 1 on purpose: the reference draws from a MULTIVARIATE t (one shared chi-square per sample,
    main_density_est.py:316-318) but weights with a PRODUCT of univariate t densities (:301), so for
    dx > 1 its estimator is not consistent; in one dimension the two coincide."""
    dx, dy, sd, scale = 1, 3, 0.5, 1.0
    A = np.array([[1.0, 0.5, -0.25]], np.float32)
    b = np.array([0.1, -0.2, 0.3], np.float32)
    ws, bs = linear_g(dx, dy, 128, 3, A, b)
    g = is_ref.DenseNet(ws, bs)
    x = np.random.RandomState(1).standard_normal((10, dx)).astype(np.float32)
    np.testing.assert_allclose(g(x), x @ A + b, rtol=1e-5, atol=1e-6)
    # proposal centred at the posterior mean makes the estimate tight
    cov = A.T.astype(np.float64) @ A + sd * sd * np.eye(dy)
    ys = np.array([[0.3, -0.1, 0.2], [1.0, 0.5, -0.5]])
    S = 20000
    og = c_oracle.OracleMlp(ws, bs)
    for y in ys:
        post_cov = np.linalg.inv(np.eye(dx) + A.astype(np.float64) @ A.T / sd ** 2)
        post_mean = post_cov @ (A.astype(np.float64) @ (y - b)) / sd ** 2
        np.random.seed(0)
        xs = is_ref.sample_from_qx(post_mean, dx, scale, 3, S)
        want = gaussian_logpdf(y, b.astype(np.float64), cov)
        got = c_oracle.is_estimate_given_samples(og, y[None], post_mean[None].astype(np.float32), xs[None], sd,
                                                 scale, 3.0, True, 0, nthreads=8)[0]
        assert abs(got - want) < 0.03, (got, want)


def test_reduction_edge_cases():
    """All-equal scores, and a far-away point whose linear-space estimate underflows to exactly 0."""
    ws, bs = glorot([2, 128, 128, 2], 3)
    og = c_oracle.OracleMlp(ws, bs)
    g = is_ref.DenseNet(ws, bs)
    h = is_ref.DenseNet(*glorot([2, 128, 128, 2], 4))
    y = np.array([[0.1, 0.2], [300.0, -300.0]])
    xhat = np.zeros((2, 2), np.float32)
    xs = np.tile(np.array([0.25, -0.5]), (2, 50, 1))     # 50 identical samples
    got = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.05, 0.5, 1.0, True, 0)
    one = c_oracle.is_estimate_given_samples(og, y, xhat, xs[:, :1], 0.05, 0.5, 1.0, True, 0)
    np.testing.assert_allclose(got, one, rtol=1e-12)
    lin = c_oracle.is_estimate_given_samples(og, y, xhat, xs, 0.05, 0.5, 1.0, False, 0)
    assert lin[1] == 0.0 and lin[0] > 0.0
    ref = is_ref.estimate_py_with_IS(g, h, y, 0.05, 0.5, 50, log=False, x_samples=xs)
    assert ref[1] == 0.0


def test_oracle_sampler_statistics():
    """The restated Philox proposal sampler (what the GPU draws are checked against): df = 1 gives
    independent-direction multivariate Cauchy; each coordinate is Cauchy(xhat_d, scale)."""
    xhat = np.array([[0.5, -1.0]], np.float32)
    xs = c_oracle.draw_samples(xhat, 0, 20000, 0.5, 1.0, seed=0)[0].astype(np.float64)
    for d in range(2):
        ks = stats.kstest((xs[:, d] - xhat[0, d]) / 0.5, "cauchy")
        assert ks.pvalue > 1e-3, ks
    # different y index / seed => different draws; same => identical
    assert np.array_equal(xs.astype(np.float32), c_oracle.draw_samples(xhat, 0, 20000, 0.5, 1.0, seed=0)[0])
    assert not np.array_equal(xs.astype(np.float32), c_oracle.draw_samples(xhat, 1, 20000, 0.5, 1.0, seed=0)[0])
    assert not np.array_equal(xs.astype(np.float32), c_oracle.draw_samples(xhat, 0, 20000, 0.5, 1.0, seed=1)[0])


def test_jacobian_restatements_agree_and_match_finite_differences():
    """get_jacobian (main_density_est.py:252-263) restated twice: numpy float32 forward mode (is_ref) and the C
    oracle in both summation orders; all agree with float64 central differences of the same MLP."""
    ws, bs = glorot([3, 128, 128, 128, 5], 31, 32)
    net = is_ref.DenseNet(ws, bs)
    om = c_oracle.OracleMlp(ws, bs)
    x = np.random.RandomState(8).standard_normal((9, 3)).astype(np.float32)
    J = is_ref.get_jacobian(net, x)
    assert J.shape == (9, 5, 3) and J.dtype == np.float64
    for order in (c_oracle.ORDER_NATURAL, c_oracle.ORDER_KERNEL):
        out, Jc = om.jacobian(x, order, nthreads=4)
        assert np.array_equal(out, om.forward(x, order))
        assert np.abs(Jc - J).max() <= 2e-6 * np.abs(J).max()

    def f64(xx):
        h = xx.astype(np.float64)
        for l, (w, b) in enumerate(zip(ws, bs)):
            h = h @ w.astype(np.float64) + b
            if l < len(ws) - 1:
                h = np.maximum(h, 0.2 * h)
        return h
    eps = 1e-6
    fd = np.stack([(f64(x + eps * np.eye(3)[d]) - f64(x - eps * np.eye(3)[d])) / (2 * eps) for d in range(3)], axis=2)
    assert np.abs(fd - J).max() <= 1e-5 * np.abs(J).max()


def test_cf_linear_generator_known_answer():
    """With G(x) = x A + b the Laplace approximation is exact whatever point it expands at, so
    estimate_py_with_CF (main_density_est.py:351-387) must give log N(y; b, A^T A + sd^2 I); the log=False branch
    carries the reference's extra sd^dy factor (:384)."""
    dx, dy, sd = 2, 3, 0.4
    A = np.array([[1.0, 0.5, -0.25], [0.3, -0.7, 0.9]], np.float32)
    b = np.array([0.1, -0.2, 0.3], np.float32)
    g = is_ref.DenseNet(*linear_g(dx, dy, 128, 3, A, b))
    h = is_ref.DenseNet(*glorot([dy, 128, 128, dx], 17, 18))      # any encoder: the expansion point is arbitrary
    ys = np.random.RandomState(2).standard_normal((6, dy))
    cov = A.T.astype(np.float64) @ A + sd * sd * np.eye(dy)
    want = np.array([gaussian_logpdf(y, b.astype(np.float64), cov) for y in ys])
    got = is_ref.estimate_py_with_CF(g, h, ys, sd_y=sd, log=True)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    lin = is_ref.estimate_py_with_CF(g, h, ys, sd_y=sd, log=False)
    np.testing.assert_allclose(np.log(lin) - dy * np.log(sd), want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", ["indep_gmm_2_2", "arem_3_6", "ypmsd_20_90", "df3_biased_2_2"])
def test_cf_golden_is_reproducible(name):
    """The committed Laplace-estimator goldens come from is_ref.estimate_py_with_CF (oracle/make_golden.py)."""
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    gw, gb, hw, hb = golden_nets(f)
    g, h = is_ref.DenseNet(gw, gb), is_ref.DenseNet(hw, hb)
    np.testing.assert_allclose(is_ref.get_jacobian(g, f["xhat"]), f["jac_at_xhat"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(is_ref.estimate_py_with_CF(g, h, f["y"], sd_y=float(f["sd"]), log=True),
                               f["cf_log_est"], rtol=1e-6)


def test_philox_known_answers():
    """The restated generator is the published Philox4x32-10 (Salmon et al., SC'11): the known-answer vectors of the
    Random123 distribution (kat_vectors: philox4x32, 10 rounds).  tests/test_gpu_parity.py asserts the same words on
    the GPU."""
    kat = [([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, out in kat:
        assert list(map(int, c_oracle.philox4x32_10(ctr, key))) == out
    # an independent statement of the round function in Python integers
    def py_philox(c, k):
        c, k = list(map(int, c)), list(map(int, k))
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        return c
    rnd = np.random.RandomState(5).randint(0, 2 ** 32, size=(200, 6), dtype=np.uint64)
    for r in rnd:
        assert list(map(int, c_oracle.philox4x32_10(r[:4], r[4:]))) == py_philox(r[:4], r[4:])


@pytest.mark.parametrize("df", [0.7, 2.5, 7.3, 40.0])
def test_oracle_gamma_path_statistics(df):
    """Non-integer df: chi2 = 2 Gamma(df/2) (Marsaglia-Tsang), restated from rt_sampler.hpp: Student-t marginals and
    the radial law of a bivariate t (r^2/2 ~ F(2, df))."""
    xs = c_oracle.draw_samples(np.zeros((1, 2), np.float32), 17, 200000, 1.0, df, 1)[0].astype(np.float64)
    for d in range(2):
        assert stats.kstest(xs[:, d], stats.t(df).cdf).pvalue > 1e-3
    assert stats.kstest((xs ** 2).sum(axis=1) / 2.0, stats.f(2, df).cdf).pvalue > 1e-3


def test_scorer_log1p_algorithm_on_the_host(tmp_path):
    """The fused kernels' scorer evaluates log(1 + a) with its own float64 routine (rt_kernels.hpp log1p_pos); the same
    operations in the same order, compiled for the host without contraction, stay within 2.5 ulp of long-double log1pl
    over 2e6 arguments in [1e-304, 1e304] and agree with libm at the corner cases (scripts/log1p_check.c exits non-zero
    otherwise).  The device function itself is checked by tests/test_gpu_parity.py::test_scorer_log1p_against_libm."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "log1p_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(root, "scripts", "log1p_check.c"), "-lm"])
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    lines = out.stdout.splitlines()
    assert "worst" in lines[0]
    for ln in lines[1:]:
        a = ln.split()[2]
        if a not in ("nan",):
            assert float(ln.split("ulp")[0].split()[-1]) <= 1.0, ln
