"""Measurement: the Laplace estimator (estimate_py_with_CF, main_density_est.py:351-387) end to end through the drop-in
class (host arrays in and out) and device-resident (rt_cf_estimate_dev), in query points per second, with the GPU share
of the wall time and the numpy restatement (oracle/is_ref.py: the reference's own per-point np.dot / inv / det) timed on
a sample beside it.  Run on the GPU box:  python scripts/cf_rate.py > gpurun_out/cf_rate.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import Generator, RoundtripModel, _lib
from oracle import is_ref

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
    # the same with the query points resident in HBM: kernels only
    eng = m.engine
    y_d, o_d = eng.to_device(y), eng.dev_alloc(n * 8)
    eng.cf_estimate_dev(y_d, n, 0.1, o_d)
    eng.sync()
    t0 = time.perf_counter()
    eng.cf_estimate_dev(y_d, n, 0.1, o_d)
    eng.sync()
    dg = time.perf_counter() - t0
    same = bool(np.array_equal(eng.from_device(o_d, (n,), np.float64), out))
    eng.dev_free(y_d); eng.dev_free(o_d)
    # CPU beside it: the statement-by-statement numpy restatement on a sample
    ns = min(n, 2000)
    gn, hn = is_ref.DenseNet(g.weights, g.biases, g.alpha), is_ref.DenseNet(h.weights, h.biases, h.alpha)
    t0 = time.perf_counter()
    ref = is_ref.estimate_py_with_CF(gn, hn, y[:ns], sd_y=0.1, log=True)
    dc = time.perf_counter() - t0
    dev = float(np.max(np.abs(out[:ns] - ref) / np.maximum(1.0, np.abs(ref))))
    # algorithmic flops per query point: G forward + dx tangent columns + H forward (the kernel recomputes the primal per
    # direction: 2 dx columns — not counted)
    jflops = 2.0 * (dx * 512 + 9 * 512 * 512 + 512 * dy) * (1 + dx) + 2.0 * (dy * 256 + 9 * 256 * 256 + 256 * dx)
    print("%-14s N = %7d: %.3f s end to end (host arrays) = %.0f points/s; device-resident %.3f s = %.0f points/s = %.1f "
          "TFLOP/s on (1 + dx) G rows + H per point; GPU share of the wall time %.0f %%; same bits: %s; finite: %s; numpy "
          "restatement on %d points: %.0f points/s, max |diff| / max(1, |log p|) = %.1e"
          % (name, n, dt, n / dt, dg, n / dg, n * jflops / dg / 1e12, 100.0 * dg / dt, same, bool(np.all(np.isfinite(out))),
             ns, ns / dc, dev), flush=True)
    m.engine.close()
