"""Diagnostic: per-phase cycle shares of the fused kernel (workgroup 0), from a -DRT_STAMPS build.
Build here:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DRT_STAMPS -o build_diag/librt_is_stamps.so roundtrip_amd/csrc/rt_is.hip
Run on the GPU box:  RT_IS_LIB=build_diag/librt_is_stamps.so python scripts/stamps.py [dx dy [g_hidden]]
"""
import ctypes as C, os, sys
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
buf = (C.c_ulonglong * 56)()
eng.is_estimate(y, xh, 6400, 0.1, 0.5)
lib.rt_debug_stamps(eng._h, buf, 1)
eng.reset_counters()
eng.is_estimate(y, xh, 6400, 0.1, 0.5)
c = eng.counters()
lib.rt_debug_stamps(eng._h, buf, 1)
# plain schedule (widths 256/1024): 2 = wait-all-read barrier, 3 = store_act, 4 = post-store barrier, 5 = hidden gemm
# skewed schedule (widths 128/512): 1 = layer 0 and the last tile-B stores, 3 = barrier P, 4 = barriers Q, R, S,
# 5 = steady stages 2 .. NKG-1, 8-11 = the four one-tile edge stages, 12 / 13 = the leaky-ReLU clusters of A / B
names = ["sample+barrier", "layer 0 + last B stores", "wait-all-read barrier (plain)",
         "barrier P (skewed) / store_act", "barriers Q R S (skewed) / post-store barrier",
         "steady MFMA stages", "last layer", "score+lse", "stage 0 (A + B's stores)", "stage 1 (A)",
         "stage NKG (B + A's stores)", "stage NKG+1 (B)", "lrelu cluster A", "lrelu cluster B"]
a = np.array(list(buf), dtype=np.float64).reshape(4, 14)
print("kernel ms", c["kernel_ms"], "TF", c["flops"] / c["kernel_ms"] / 1e9)
for w in range(4):
    tot = a[w].sum()
    if tot == 0:
        continue            # one- / two-wave workgroups
    print("wave %d total %.0f ticks: " % (w, tot) + ", ".join("%s %.2f%%" % (n, 100 * v / tot) for n, v in zip(names, a[w])))
