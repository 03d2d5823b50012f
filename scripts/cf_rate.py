"""Measurement: the Laplace estimator (estimate_py_with_CF, main_density_est.py:351-387) end to end — H forward, G forward
and Jacobian on the GPU, the x_dim x x_dim algebra in float64 numpy — in query points per second, with the GPU share.
Run on the GPU box:  python scripts/cf_rate.py > gpurun_out/cf_rate.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import Generator, RoundtripModel, _lib

for name, dx, dy, n in (("indep_gmm 2/2", 2, 2, 1000000), ("AReM 3/6", 3, 6, 1000000), ("HEPMASS 8/21", 8, 21, 200000),
                        ("YPMSD 20/90", 20, 90, 100000)):
    rng = np.random.RandomState(1234)
    g = Generator(dx, dy, "g_net", nb_layers=10, nb_units=512, rng=rng)
    h = Generator(dy, dx, "h_net", nb_layers=10, nb_units=256, rng=rng)
    m = RoundtripModel(g, h, data="cf", df=1, device=0)
    y = np.random.RandomState(42).standard_normal((n, dy)).astype(np.float32).astype(np.float64)
    m.estimate_py_with_CF(y[:4096], 0, sd_y=0.1, save=False)              # warm-up: code objects, buffers
    t0 = time.perf_counter()
    out = m.estimate_py_with_CF(y, 0, sd_y=0.1, save=False)
    dt = time.perf_counter() - t0
    # the GPU part alone on the same points (H forward, then G forward + Jacobian), host arrays in and out
    t0 = time.perf_counter()
    cb = max(1, (64 << 20) // (dx * dy * 8))
    for i in range(0, n, cb):
        x_ = m.engine.mlp_forward(_lib.RT_MLP_H, y[i:i + cb].astype(np.float32))
        m.engine.mlp_jacobian(_lib.RT_MLP_G, x_)
    dg = time.perf_counter() - t0
    jflops = 2.0 * (dx * 512 + 9 * 512 * 512 + 512 * dy) * (1 + dx) + 2.0 * (dy * 256 + 9 * 256 * 256 + 256 * dx)
    print("%-14s N = %7d: %.2f s end to end = %.0f points/s; GPU calls (incl. PCIe, J = %d MB) %.2f s = %.1f TFLOP/s on "
          "(1 + dx) G rows + H per point; finite: %s" % (name, n, dt, n / dt, n * dx * dy * 4 >> 20, dg, n * jflops / dg / 1e12,
                                                         bool(np.all(np.isfinite(out)))), flush=True)
    m.engine.close()
