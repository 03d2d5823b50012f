"""Diagnostic: per-phase cycle shares of the fused kernel (workgroup 0), from a -DRT_STAMPS build.
Build here:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DRT_STAMPS -o build_diag/librt_is_stamps.so roundtrip_amd/csrc/rt_is.hip
Run on the GPU box:  RT_IS_LIB=build_diag/librt_is_stamps.so python scripts/stamps.py [dx dy]
"""
import ctypes as C, os, sys
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
buf = (C.c_ulonglong * 32)()
eng.is_estimate(y, xh, 6400, 0.1, 0.5)
lib.rt_debug_stamps(eng._h, buf, 1)
eng.reset_counters()
eng.is_estimate(y, xh, 6400, 0.1, 0.5)
c = eng.counters()
lib.rt_debug_stamps(eng._h, buf, 1)
# plain schedule (widths 256/1024): 2 = wait-all-read barrier, 3 = store_act, 4 = post-store barrier, 5 = hidden gemm
# skewed schedule (widths 128/512):  1 also holds the first / last half-epilogues, 2 = head stages 0-1 (A MFMAs +
# B's epilogue), 3 = barrier before stage 2, 4 = the two barriers around stage NKG+1, 5 = stages 2 .. NKG+1
names = ["sample+barrier", "layer0 + unhidden half-epilogues", "head stages (skewed) / wait-all-read barrier",
         "barrier after head (skewed) / store_act", "tail barriers (skewed) / post-store barrier",
         "hidden MFMA stages", "last layer", "score+lse"]
a = np.array(list(buf), dtype=np.float64).reshape(4, 8)
print("kernel ms", c["kernel_ms"], "TF", c["flops"] / c["kernel_ms"] / 1e9)
for w in range(4):
    tot = a[w].sum()
    print("wave %d total %.0f ticks: " % (w, tot) + ", ".join("%s %.2f%%" % (n, 100 * v / tot) for n, v in zip(names, a[w])))
