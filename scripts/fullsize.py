"""BASELINE config 3 at its real per-GPU size: one rank's share (1/8) of 1 M AReM-shaped query points x S = 40000 through the
drop-in class, end to end from host arrays (H forward, copies, 19 bounded launches, finalize).  Checks the size-independent
properties: finite everywhere, and a handful of points re-evaluated alone (with their global index) give the same bits.
Run on the GPU box:  python scripts/fullsize.py [n_points]  > gpurun_out/fullsize.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import Generator, RoundtripModel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
dx, dy, S = 3, 6, 40000
rng = np.random.RandomState(1234)
g = Generator(dx, dy, "g_net", nb_layers=10, nb_units=512, rng=rng)
h = Generator(dy, dx, "h_net", nb_layers=10, nb_units=256, rng=rng)
m = RoundtripModel(g, h, data="uci_AReM", df=1, device=0)
y = np.random.RandomState(42).standard_normal((N, dy)).astype(np.float32).astype(np.float64)
m.estimate_py_with_IS(y[:256], 0, sd_y=0.1, scale=0.1, sample_size=2000, save=False)      # warm-up
m.engine.reset_counters()
t0 = time.perf_counter()
out = m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=S, save=False)
dt = time.perf_counter() - t0
c = m.engine.counters()
idx = [0, 1, N // 3, N - 1]
alone = np.array([m.engine.is_estimate(y[i:i + 1], m.engine.mlp_forward(1, y[i:i + 1].astype(np.float32)), S, 0.1, 0.1,
                                      df=1, seed=m.is_seed, log=True, y_index_base=i)[0] for i in idx])
print(json.dumps({"config": "AReM 3/6, one rank's share of config 3", "N": N, "S": S, "seconds_end_to_end": dt,
                  "sample_evals_per_s_end_to_end": N * S / dt, "launches": c["launches"],
                  "kernel_tflops": c["flops"] / (c["kernel_ms"] * 1e-3) / 1e12,
                  "all_finite": bool(np.isfinite(out).all()), "mean_log_p": float(out.mean()),
                  "points_alone_bit_identical": bool(np.array_equal(alone, out[idx]))}))
