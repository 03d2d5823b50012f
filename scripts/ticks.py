"""Diagnostic A/B measure that does not depend on the box's clock: total s_memtime ticks of the fused kernel (workgroup 0,
slowest wave) from a -DRT_STAMPS=2 build, per 64-sample pass, against the cycles of pure MFMA time at the unpadded
algorithmic flops (1 180 672 at dx = dy = 2, G = 10 x 512).  RT_IS_LIB=build_diag/<lib>.so python scripts/ticks.py [dx dy]"""
import ctypes as C, os, sys
os.environ["RT_IS_BALANCE"] = "0"      # equal shares: workgroup 0 runs exactly 800 passes (read by rt_ctx_create)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import _lib, Generator
dx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dy = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.RandomState(1234)
g = Generator(dx, dy, "g", 10, 512, rng=rng); h = Generator(dy, dx, "h", 10, 256, rng=rng)
eng = _lib.Engine(0)
eng.set_mlp(0, g.weights, g.biases); eng.set_mlp(1, h.weights, h.biases)
y = np.random.RandomState(1).standard_normal((2048, dy)); xh = eng.mlp_forward(1, y.astype(np.float32))
lib = _lib.load()
buf = (C.c_ulonglong * 56)()
res = []
for rep in range(3):
    eng.is_estimate(y, xh, 6400, 0.1, 0.5)
    lib.rt_debug_stamps(eng._h, buf, 1)
    t = max(buf[w * 14] for w in range(4)) / 800.0       # 2048 points x 100 passes / 256 workgroups
    res.append(t)
ideal = 2.0 * (dx * 512 + 9 * 512 * 512 + 512 * dy) * 64 / 256.0      # cycles of a CU's 256 flop/clk on one pass
print("%s %d/%d: ticks per pass %s -> %.2f %% of the MFMA issue rate" % (os.environ.get("RT_IS_LIB"), dx, dy,
                                                                     ["%.0f" % r for r in res], 100 * ideal / min(res)))
