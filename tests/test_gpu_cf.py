"""GPU tests of the Laplace (closed-form) estimator on the device (rt_cf_estimate: estimate_py_with_CF,
main_density_est.py:351-387) and of the device-side mean / standard error (rt_mean_stderr: :187-188), through the C ABI.

Tolerances: the float64 algebra against the reference's formulas evaluated by numpy ON THE SAME Jacobian (the C oracle's,
kernel summation order, bit-identical to the GPU's): 1e-9 relative; against the statement-by-statement numpy restatement
of the whole estimator (natural summation order, float32 rounding of G and J differs): 1e-4 relative (north_star);
mean / standard error against numpy: the same bits."""
import os

import numpy as np
import pytest

from oracle import c_oracle, is_ref
from roundtrip_amd import _lib

from test_oracle import GOLDEN, golden_nets

pytestmark = pytest.mark.gpu

G, H = _lib.RT_MLP_G, _lib.RT_MLP_H
DIMS = [(2, 2), (3, 6), (5, 9), (8, 17), (8, 21), (20, 90)]


def cf_reference_formulas(y, x_, y__, J, sd, log):
    """main_density_est.py:365-384 literally (np.dot, np.linalg.inv, np.linalg.det), point by point, from given
    network outputs."""
    dx, dy = x_.shape[1], y.shape[1]
    out = []
    for n in range(len(y)):
        r = y[n] - y__[n]
        A = np.dot(J[n].T, J[n])
        b = np.dot(J[n].T, r)
        Sigma = np.linalg.inv(np.eye(dx) + A / sd ** 2)
        Sigma_inv = np.eye(dx) + A / sd ** 2
        mu = Sigma.dot(b / sd ** 2 - x_[n])
        c = np.sum(x_[n] ** 2) + np.sum(r ** 2) / sd ** 2 - mu.T.dot(Sigma_inv).dot(mu)
        if log:
            out.append(-dy * np.log(np.sqrt(2 * np.pi) * sd) + 0.5 * np.log(np.linalg.det(Sigma)) - 0.5 * c)
        else:
            out.append(1. / (np.sqrt(2 * np.pi) * sd) ** dy * sd ** dy * np.sqrt(np.linalg.det(Sigma)) * np.exp(-0.5 * c))
    return np.array(out)


@pytest.mark.parametrize("dx,dy", DIMS)
def test_cf_estimate_against_the_reference_formulas(engine, make_nets, dx, dy):
    gw, gb, hw, hb = make_nets(dx, dy)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    og, oh = c_oracle.OracleMlp(gw, gb), c_oracle.OracleMlp(hw, hb)
    rng = np.random.RandomState(31 * dx + dy)
    y = rng.standard_normal((77, dy)) * 0.4
    x_ = oh.forward(y.astype(np.float32), c_oracle.ORDER_KERNEL, nthreads=8).astype(np.float64)
    y__, J = og.jacobian(x_, c_oracle.ORDER_KERNEL, nthreads=8)
    y__, J = y__.astype(np.float64), J.astype(np.float64)
    for sd in (0.5, 0.1):
        got = engine.cf_estimate(y, sd, log=True)
        assert got.shape == (77,) and got.dtype == np.float64 and np.all(np.isfinite(got))
        np.testing.assert_allclose(got, cf_reference_formulas(y, x_, y__, J, sd, True), rtol=1e-9)
        lin = engine.cf_estimate(y, sd, log=False)
        want_lin = cf_reference_formulas(y, x_, y__, J, sd, False)
        ok = want_lin > 1e-300
        np.testing.assert_allclose(lin[ok], want_lin[ok], rtol=1e-9)
        assert np.all(lin[~ok] <= 1e-300)
        # the whole estimator restated in numpy (natural summation order): north_star tolerance
        # (log p crosses zero: the absolute term covers values near it — the float32 rounding of G and J enters
        # through 1 / sd^2)
        want = is_ref.estimate_py_with_CF(is_ref.DenseNet(gw, gb), is_ref.DenseNet(hw, hb), y, sd_y=sd, log=True)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)


def test_cf_estimate_dev_blocks_and_edges(make_nets, monkeypatch):
    """Device pointers; a call split into blocks of 32 / 64 points gives the one-block bits; N = 0, 1, 33; errors."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    gw, gb, hw, hb = make_nets(5, 9)
    rng = np.random.RandomState(9)
    y = rng.standard_normal((150, 9)) * 0.5
    res = {}
    for blk in (None, "32", "64"):
        if blk:
            monkeypatch.setenv("RT_CF_BLOCK_POINTS", blk)
        eng = _lib.Engine(0)
        try:
            eng.set_mlp(G, gw, gb)
            with pytest.raises(_lib.RtError):
                eng.cf_estimate(y, 0.1)                                  # H missing
            eng.set_mlp(H, hw, hb)
            y_d = eng.to_device(y)
            o_d = eng.to_device(np.full(150, -1.0))
            eng.cf_estimate_dev(y_d, 150, 0.1, o_d, log=True)
            eng.sync()
            res[blk] = eng.from_device(o_d, (150,), np.float64)
            assert np.array_equal(res[blk], eng.cf_estimate(y, 0.1))
            for n in (1, 33):
                assert np.array_equal(eng.cf_estimate(y[:n], 0.1), res[blk][:n])
            assert eng.cf_estimate(np.zeros((0, 9)), 0.1).shape == (0,)
            with pytest.raises(_lib.RtError):
                eng.cf_estimate(y, -1.0)
            with pytest.raises(ValueError):
                eng.cf_estimate(y[:, :4], 0.1)
            eng.dev_free(y_d)
            eng.dev_free(o_d)
        finally:
            eng.close()
    assert np.array_equal(res[None], res["32"]) and np.array_equal(res[None], res["64"])


@pytest.mark.parametrize("dx,dy,hid", [(2, 2, 512), (3, 6, 512), (5, 9, 512), (8, 21, 512), (20, 90, 512), (8, 21, 64),
                                       (5, 9, 128), (3, 6, 256), (2, 2, 64), (4, 4, 128), (2, 3, 256), (6, 2, 256)])
def test_jacobian_with_slope_masks_has_the_same_bits(make_nets, monkeypatch, dx, dy, hid):
    """mlp_jvp_first_kernel + mlp_jvp_rest_kernel (the primal once per block of 32 rows, its leaky-ReLU slopes kept as bit masks for the tangent-only
    passes that follow, two (block of rows, direction) items per pass — for an even dx a pass straddles two blocks)
    against mlp_jvp_kernel (the primal recomputed per direction) and the C oracle's forward mode in
    kernel order: G(x) and J bit for bit, on ragged row counts; and the Laplace estimate built on either."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    from test_gpu_parity import glorot_nets
    if hid == 512:
        gw, gb, hw, hb = make_nets(dx, dy)
    else:
        gw, gb = glorot_nets(dx, hid, dy, 7, n_hidden=5)
        hw, hb = make_nets(dx, dy)[2:]
    engines = {}
    try:
        for mode in ("0", "2"):
            monkeypatch.setenv("RT_JVP_MASKS", mode)
            engines[mode] = _lib.Engine(0)
            engines[mode].set_mlp(G, gw, gb)
            engines[mode].set_mlp(H, hw, hb)
        og = c_oracle.OracleMlp(gw, gb)
        rng = np.random.RandomState(dx + dy)
        for n in (1, 31, 32, 33, 65, 100):
            x = (rng.standard_normal((n, dx)) * 1.5).astype(np.float32)
            o0, j0 = engines["0"].mlp_jacobian(G, x)
            o2, j2 = engines["2"].mlp_jacobian(G, x)
            assert np.array_equal(o0, o2) and np.array_equal(j0, j2), n
            if n <= 33:
                wo, wj = og.jacobian(x, c_oracle.ORDER_KERNEL, nthreads=8)
                assert np.array_equal(o2, wo) and np.array_equal(j2, wj), n
        y = rng.standard_normal((150, dy)) * 0.4
        assert np.array_equal(engines["0"].cf_estimate(y, 0.1), engines["2"].cf_estimate(y, 0.1))
    finally:
        for e in engines.values():
            e.close()


@pytest.mark.parametrize("dx,dy", [(5, 9), (2, 2)])
def test_jacobian_kernel_choice_on_a_large_call(make_nets, monkeypatch, dx, dy):
    """The default picks the mask kernels where they leave the busiest workgroup fewer passes (many rows, dx >= 2): 20 000
    rows at 5/9 and 2/2 — same bits as the per-direction kernel, and as the oracle on a sample of the rows."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    gw, gb, hw, hb = make_nets(dx, dy)
    x = (np.random.RandomState(1).standard_normal((20000, dx)) * 1.5).astype(np.float32)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RT_JVP_MASKS", mode)
        eng = _lib.Engine(0)
        try:
            eng.set_mlp(G, gw, gb)
            res[mode] = eng.mlp_jacobian(G, x)
        finally:
            eng.close()
    assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
    pick = np.random.RandomState(2).choice(20000, 40, replace=False)
    wo, wj = c_oracle.OracleMlp(gw, gb).jacobian(x[pick], c_oracle.ORDER_KERNEL, nthreads=8)
    assert np.array_equal(res["1"][0][pick], wo) and np.array_equal(res["1"][1][pick], wj)


@pytest.mark.parametrize("case_seed", range(24))
def test_cf_random_shapes_against_the_reference_formulas(engine, case_seed):
    """Random generator / encoder shapes (dims 1..40, hidden widths 16..512 in any mix, 2..6 layers, sd 0.03..2, ragged
    N): rt_cf_estimate against the reference's formulas on the C oracle's kernel-order network outputs."""
    rng = np.random.RandomState(1000 + case_seed)
    dx, dy = int(rng.randint(1, 41)), int(rng.randint(1, 41))
    gh = [int(rng.choice([16, 50, 64, 100, 128, 200, 256, 300, 512])) for _ in range(rng.randint(1, 6))]
    hh = [int(rng.choice([32, 64, 128, 256])) for _ in range(rng.randint(1, 4))]
    gw, gb = is_ref.glorot_mlp([dx] + gh + [dy], rng)
    gb = [rng.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in gb]
    hw, hb = is_ref.glorot_mlp([dy] + hh + [dx], rng)
    sd = float(10.0 ** rng.uniform(-1.5, 0.3))
    n = int(rng.choice([1, 5, 32, 33, 70]))
    try:
        engine.set_mlp(G, gw, gb)
    except _lib.RtError as e:
        assert e.code == 4 and max(gh) <= 1024          # RT_EUNSUPPORTED: the LDS budget, with its own test elsewhere
        pytest.skip("shape outside the LDS budget: %s" % e)
    engine.set_mlp(H, hw, hb)
    y = rng.standard_normal((n, dy)) * 0.5
    og, oh = c_oracle.OracleMlp(gw, gb), c_oracle.OracleMlp(hw, hb)
    x_ = oh.forward(y.astype(np.float32), c_oracle.ORDER_KERNEL, nthreads=8).astype(np.float64)
    y__, J = og.jacobian(x_, c_oracle.ORDER_KERNEL, nthreads=8)
    want = cf_reference_formulas(y, x_, y__.astype(np.float64), J.astype(np.float64), sd, True)
    got = engine.cf_estimate(y, sd, log=True)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9, err_msg=str((dx, dy, gh, hh, sd, n)))


@pytest.mark.parametrize("dx,dy,hid", [(1, 1, 128), (1, 7, 512), (64, 70, 256), (100, 128, 512), (128, 30, 256)])
def test_cf_extreme_dims(engine, dx, dy, hid):
    """The smallest dims and the largest the kernels take (dx = 128: a 128 x 128 float64 factorisation in 128 KB of LDS,
    two rows per lane in the column loops)."""
    from test_gpu_parity import glorot_nets
    gw, gb = glorot_nets(dx, hid, dy, 3, n_hidden=2)
    hw, hb = glorot_nets(dy, 128, dx, 4, n_hidden=2)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    rng = np.random.RandomState(dx)
    y = rng.standard_normal((9, dy)) * 0.5
    og, oh = c_oracle.OracleMlp(gw, gb), c_oracle.OracleMlp(hw, hb)
    x_ = oh.forward(y.astype(np.float32), c_oracle.ORDER_KERNEL, nthreads=8).astype(np.float64)
    y__, J = og.jacobian(x_, c_oracle.ORDER_KERNEL, nthreads=8)
    for sd in (0.5, 0.05):
        want = cf_reference_formulas(y, x_, y__.astype(np.float64), J.astype(np.float64), sd, True)
        np.testing.assert_allclose(engine.cf_estimate(y, sd, log=True), want, rtol=1e-9, atol=1e-9)


def test_cf_full_size_properties(engine, make_nets):
    """BASELINE-size call (200 000 HEPMASS-shaped points: several blocks of the Jacobian buffer, the slope-mask kernel
    on every one of them) through size-independent properties: every result finite; 300 points picked at random and
    evaluated alone have the same bits; log and linear agree through the reference's extra sd^dy; the device-side mean and
    standard error of the results have numpy's bits."""
    gw, gb, hw, hb = make_nets(8, 21)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    n, sd = 200000, 0.5
    y = np.random.RandomState(42).standard_normal((n, 21)).astype(np.float32).astype(np.float64)
    got = engine.cf_estimate(y, sd, log=True)
    assert got.shape == (n,) and np.all(np.isfinite(got))
    pick = np.sort(np.random.RandomState(1).choice(n, 300, replace=False))
    assert np.array_equal(engine.cf_estimate(y[pick], sd, log=True), got[pick])
    lin = engine.cf_estimate(y[:5000], sd, log=False)
    ok = lin > 1e-300
    np.testing.assert_allclose(np.log(lin[ok]) - 21 * np.log(sd), got[:5000][ok], rtol=1e-9, atol=1e-9)
    mean, se = engine.mean_stderr(got)
    assert mean == np.mean(got) and se == np.std(got) / np.sqrt(n)


@pytest.mark.parametrize("name", ["indep_gmm_2_2", "arem_3_6", "ypmsd_20_90", "df3_biased_2_2"])
def test_cf_estimate_golden(engine, name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    gw, gb, hw, hb = golden_nets(f)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    np.testing.assert_allclose(engine.cf_estimate(f["y"], float(f["sd"]), log=True), f["cf_log_est"], rtol=1e-4)


def test_cf_linear_generator_known_answer_on_gpu(engine):
    """A generator built to be exactly linear, G(x) = x A + b: the Laplace approximation is then exact whatever point
    it expands at, p(y) = N(y; b, A^T A + sd^2 I) (tests/test_oracle.py::test_cf_linear_generator_known_answer states
    the same for the CPU restatement); the log=False branch carries the reference's extra sd^dy factor (:384)."""
    from test_oracle import gaussian_logpdf, linear_g
    from conftest import glorot
    dx, dy, sd = 2, 3, 0.4
    A = np.array([[1.0, 0.5, -0.25], [0.3, -0.7, 0.9]], np.float32)
    b = np.array([0.1, -0.2, 0.3], np.float32)
    gw, gb = linear_g(dx, dy, 128, 3, A, b)
    hw, hb = glorot([dy, 128, 128, dx], 17, 18)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    ys = np.random.RandomState(2).standard_normal((40, dy))
    cov = A.T.astype(np.float64) @ A + sd * sd * np.eye(dy)
    want = np.array([gaussian_logpdf(y, b.astype(np.float64), cov) for y in ys])
    got = engine.cf_estimate(ys, sd, log=True)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    lin = engine.cf_estimate(ys, sd, log=False)
    np.testing.assert_allclose(np.log(lin) - dy * np.log(sd), got, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n", [0, 1, 5, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 100003, 1000000])
def test_mean_stderr_has_numpys_bits(engine, n):
    rng = np.random.RandomState(n + 1)
    v = rng.standard_normal(n) * rng.uniform(0.1, 50.0) - 300.0
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_mean = np.mean(v)
            want_se = np.std(v) / np.sqrt(len(v)) if n else np.float64("nan")
    mean, se = engine.mean_stderr(v)
    if n == 0:
        assert np.isnan(mean) and np.isnan(se)
        return
    assert mean == want_mean and se == want_se, (mean, want_mean, se, want_se)
    # device-resident variant
    v_d = engine.to_device(v)
    o_d = engine.dev_alloc(16)
    engine.mean_stderr_dev(v_d, n, o_d)
    engine.sync()
    got = engine.from_device(o_d, (2,), np.float64)
    engine.dev_free(v_d)
    engine.dev_free(o_d)
    assert got[0] == want_mean and got[1] == want_se


def test_mean_stderr_many_small_arrays(engine):
    """600 arrays of 2..40 values: the final square roots and the division are correctly rounded on the device (its
    plain sqrt() is not: it was 2 ulp off at n = 7)."""
    rng = np.random.RandomState(99)
    for k in range(600):
        n = 2 + k % 39
        v = rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3) + rng.uniform(-100, 100)
        mean, se = engine.mean_stderr(v)
        assert mean == np.mean(v) and se == np.std(v) / np.sqrt(len(v)), (k, n)


def test_is_estimate_summary(engine, make_nets):
    gw, gb, hw, hb = make_nets(3, 6)
    engine.set_mlp(G, gw, gb)
    engine.set_mlp(H, hw, hb)
    y = np.random.RandomState(5).standard_normal((300, 6)) * 0.3
    xhat = engine.mlp_forward(H, y.astype(np.float32))
    est, mean, se = engine.is_estimate_summary(y, xhat, 500, 0.1, 0.1, df=1.0, seed=3, y_index_base=11)
    assert np.array_equal(est, engine.is_estimate(y, xhat, 500, 0.1, 0.1, df=1.0, seed=3, log=True, y_index_base=11))
    assert mean == np.mean(est) and se == np.std(est) / np.sqrt(len(est))
