"""Diagnostic A/B measure that does not depend on the box's clock: total s_memtime ticks of the fused kernel (workgroup 0,
slowest wave) from a -DRT_STAMPS=2 build, per 64-sample pass, against the cycles of pure MFMA time at the unpadded
algorithmic flops (1 180 672 at dx = dy = 2, G = 10 x 512).  RT_IS_LIB=build_diag/<lib>.so python scripts/ticks.py [dx dy [g_units]]
(g_units 64 / 128 run as one-wave workgroups, four to a CU, 256 as two-wave workgroups: a pass then has one / two SIMDs.)"""
import ctypes as C, os, sys
os.environ["RT_IS_BALANCE"] = "0"      # equal shares: workgroup 0 runs exactly 800 passes (read by rt_ctx_create)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import _lib, Generator
dx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dy = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hid = int(sys.argv[3]) if len(sys.argv) > 3 else 512
rng = np.random.RandomState(1234)
g = Generator(dx, dy, "g", 10, hid, rng=rng); h = Generator(dy, dx, "h", 10, 256, rng=rng)
eng = _lib.Engine(0)
eng.set_mlp(0, g.weights, g.biases); eng.set_mlp(1, h.weights, h.biases)
y = np.random.RandomState(1).standard_normal((2048, dy)); xh = eng.mlp_forward(1, y.astype(np.float32))
lib = _lib.load()
nw = int(eng.is_kernel_name(1.0).rstrip(">").split(",")[-1])      # waves per workgroup of the kernel this net runs on
passes = 2048 * 100 / (256 * 4 // nw)                              # 2048 points x 100 passes over 256 CUs x (4 / nw) workgroups
buf = (C.c_ulonglong * 56)()
res = []
for rep in range(3):
    eng.is_estimate(y, xh, 6400, 0.1, 0.5)
    lib.rt_debug_stamps(eng._h, buf, 1)
    t = max(buf[w * 14] for w in range(nw)) / passes
    res.append(t)
ideal = 2.0 * (dx * hid + 9 * hid * hid + hid * dy) * 64 / (64.0 * nw)   # cycles of nw SIMDs' 64 flop/clk each on one pass
print("%s %d/%d x %d: ticks per pass %s -> %.2f %% of the MFMA issue rate" % (os.environ.get("RT_IS_LIB"), dx, dy, hid,
                                                                          ["%.0f" % r for r in res], 100 * ideal / min(res)))
