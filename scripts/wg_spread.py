"""Diagnostic: how evenly the persistent workgroups of the fused kernel finish (a -DRT_STAMPS=2 build): per workgroup its
own cycle count (s_memtime) and its end time on the 100 MHz clock all CUs share.  Bench workload by default.
RT_IS_LIB=build_diag/librt_is_ticks.so python scripts/wg_spread.py [grid S]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import _lib, Generator
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 100
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.RandomState(1234)
g = Generator(2, 2, "g", 10, 512, rng=rng); h = Generator(2, 2, "h", 10, 256, rng=rng)
eng = _lib.Engine(0)
eng.set_mlp(0, g.weights, g.biases); eng.set_mlp(1, h.weights, h.biases)
v1, v2 = np.meshgrid(np.linspace(-5, 5, grid), np.linspace(-5, 5, grid))
y = np.vstack((v1.ravel(), v2.ravel())).T
xh = eng.mlp_forward(1, y.astype(np.float32))
lib = _lib.load()
buf = (C.c_ulonglong * 2048)()
for rep in range(4):
    eng.reset_counters()
    eng.is_estimate(y, xh, S, 0.1, 0.5, log=False)
    c = eng.counters()
    lib.rt_debug_wg_ticks(eng._h, buf)
    a = np.array(list(buf), dtype=np.float64).reshape(1024, 2)[:256]
    t, end = a[:, 0], a[:, 1]
    print("kernel %.2f ms; per-WG cycles: min %.4e mean %.4e max %.4e (max/mean %.4f, max/min %.4f)"
          % (c["kernel_ms"], t.min(), t.mean(), t.max(), t.max() / t.mean(), t.max() / t.min()))
    e = (end - end.min()) / 100e6 * 1e3
    print("  end times (ms after the first WG to finish): median %.2f  p90 %.2f  max %.2f" % (np.median(e), np.percentile(e, 90), e.max()))
    ex = e.reshape(32, 8)
    print("  end time by blockIdx %% 8 (ms): mean", ["%.1f" % v for v in ex.mean(axis=0)], " spread within class", ["%.1f" % v for v in (ex.max(axis=0) - ex.min(axis=0))])
    x = t.reshape(32, 8)   # blockIdx % 8 = XCD (round-robin dispatch)
    print("  mean cycles by blockIdx %% 8:", ["%.4e" % v for v in x.mean(axis=0)])
    print("  shares for the next launch (x 8):", ["%.4f" % (8 * v) for v in eng.xcd_shares()])
    print("  implied clock by WG (cycles / kernel time), GHz: min %.4f max %.4f" % (t.min() / c["kernel_ms"] / 1e6, t.max() / c["kernel_ms"] / 1e6))
