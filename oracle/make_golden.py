"""Generates tests/golden/*.npz with the numpy/scipy restatement (oracle/is_ref.py).

Run in the build container:  python -m oracle.make_golden
The reference's TensorFlow model cannot be imported here, so the vectors come from is_ref.py, which
uses the reference's own third-party arithmetic (numpy legacy RNG, scipy.stats.t).  The one piece of the
reference that IS importable without TensorFlow — src/pyroundtrip/util.py — supplies the evaluation grid
(create_2d_grid_data, util.py:624-629) when /root/reference is present, pinning our restatement of it.
Weights are not stored: tests regenerate them from the recorded RandomState seeds.
"""
import importlib.util
import os

import numpy as np

from . import is_ref

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF_UTIL = "/root/reference/src/pyroundtrip/util.py"


def reference_grid(*args, **kw):
    if os.path.exists(REF_UTIL):
        spec = importlib.util.spec_from_file_location("ref_util", REF_UTIL)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        grid = mod.create_2d_grid_data(*args, **kw)[2]
        assert np.array_equal(grid, is_ref.create_2d_grid_data(*args, **kw)[2])
        return grid, True
    return is_ref.create_2d_grid_data(*args, **kw)[2], False


def nets(dx, dy, seed, bias_seed=None):
    rng = np.random.RandomState(seed)
    gw, gb = is_ref.glorot_mlp([dx] + [512] * 10 + [dy], rng)
    hw, hb = is_ref.glorot_mlp([dy] + [256] * 10 + [dx], rng)
    if bias_seed is not None:
        r = np.random.RandomState(bias_seed)
        gb = [r.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in gb]
        hb = [r.uniform(-0.1, 0.1, size=b.shape).astype(np.float32) for b in hb]
    return is_ref.DenseNet(gw, gb), is_ref.DenseNet(hw, hb)


def make(name, dx, dy, y, sd, scale, S, df=1, seed=1234, bias_seed=None, extra=None):
    g, h = nets(dx, dy, seed, bias_seed)
    log_est, xs, xhat = is_ref.estimate_py_with_IS(g, h, y, sd_y=sd, scale=scale, sample_size=S, df=df,
                                                   log=True, return_samples=True)
    lin_est = is_ref.estimate_py_with_IS(g, h, y, sd_y=sd, scale=scale, sample_size=S, df=df, log=False,
                                         x_samples=xs)
    d = dict(dx=dx, dy=dy, weight_seed=seed, bias_seed=-1 if bias_seed is None else bias_seed, y=y,
             xhat=xhat.astype(np.float32), sd=sd, scale=scale, S=S, df=df, log_est=log_est, lin_est=lin_est,
             xs_head=xs[:, :8, :].copy(), g_of_xhat=g(xhat.astype(np.float32)),
             # Laplace estimator (main_density_est.py:351-387) and the Jacobian it uses (:252-263)
             jac_at_xhat=is_ref.get_jacobian(g, xhat),
             cf_log_est=is_ref.estimate_py_with_CF(g, h, y, sd_y=sd, log=True),
             cf_lin_est=is_ref.estimate_py_with_CF(g, h, y, sd_y=sd, log=False))
    if extra:
        d.update(extra)
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, log_est)


def main():
    os.makedirs(OUT, exist_ok=True)
    grid, from_ref = reference_grid(-1.5, 1.5, -1.5, 1.5, n=100)
    # BASELINE config 1 (plumbing): indep_gmm grid, sd .05, scale .5 (main_density_est.py:457)
    make("indep_gmm_2_2", 2, 2, grid[::1237][:8].copy(), 0.05, 0.5, 4000,
         extra=dict(grid_from_reference_util=from_ref))
    # BASELINE config 3: AReM 3/6, y ~ N(0, I) RandomState(42), sd .1 scale .1
    make("arem_3_6", 3, 6, np.random.RandomState(42).standard_normal((4, 6)), 0.1, 0.1, 2000)
    # BASELINE config 5: YearPredictionMSD 20/90
    make("ypmsd_20_90", 20, 90, np.random.RandomState(42).standard_normal((3, 90)), 0.1, 0.1, 1000)
    # general df, non-zero biases
    make("df3_biased_2_2", 2, 2, np.random.RandomState(7).standard_normal((4, 2)), 0.5, 0.5, 2000, df=3,
         bias_seed=11)


if __name__ == "__main__":
    main()
