"""Diagnostic A/B measure that does not depend on the box's clock: total s_memtime ticks of the fused kernel (workgroup 0,
slowest wave) from a -DRT_STAMPS=2 build, per 64-sample pass, against the 1 180 672 cycles of pure MFMA time
(dx = dy = 2, G = 10 x 512).  RT_IS_LIB=build_diag/<lib>.so python scripts/ticks.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import _lib, Generator
rng = np.random.RandomState(1234)
g = Generator(2, 2, "g", 10, 512, rng=rng); h = Generator(2, 2, "h", 10, 256, rng=rng)
eng = _lib.Engine(0)
eng.set_mlp(0, g.weights, g.biases); eng.set_mlp(1, h.weights, h.biases)
y = np.random.RandomState(1).standard_normal((2048, 2)); xh = eng.mlp_forward(1, y.astype(np.float32))
lib = _lib.load()
buf = (C.c_ulonglong * 56)()
res = []
for rep in range(3):
    eng.is_estimate(y, xh, 6400, 0.1, 0.5)
    lib.rt_debug_stamps(eng._h, buf, 1)
    t = max(buf[w * 14] for w in range(4)) / 800.0       # 2048 points x 100 passes / 256 workgroups
    res.append(t)
print("%s: ticks per pass %s -> %.2f %% of the MFMA issue rate" % (os.environ.get("RT_IS_LIB"), ["%.0f" % r for r in res],
                                                               100 * 1180672.0 / min(res)))
