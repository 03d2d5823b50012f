"""Throughput of the fused IS kernel on the other BASELINE.json configurations, one MI355X.
Not bench lines: parity for these shapes is in tests/; this records achieved TFLOP/s (unpadded algorithmic flops)
per shape.  Run on the GPU box:  python scripts/sweep.py > gpurun_out/sweep.json
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd import _lib, Generator

PEAK = 157.3
CONFIGS = [
    # name, dx, dy, hidden, S, sd, scale, target sample-evals per launch
    ("indep_gmm 2/2 S=40000 (config 1 shape)", 2, 2, 512, 40000, 0.05, 0.5, 1.0e8),
    ("octagon/involute 2/2 S=10000 (config 2, the bench line)", 2, 2, 512, 10000, 0.1, 0.5, 1.0e8),
    ("AReM 3/6 S=40000 (config 3)", 3, 6, 512, 40000, 0.1, 0.1, 1.0e8),
    ("CASP 5/9 S=40000 (config 3)", 5, 9, 512, 40000, 0.1, 0.1, 1.0e8),
    ("BANK 8/17 S=40000 (config 4)", 8, 17, 512, 40000, 0.1, 0.1, 1.0e8),
    ("HEPMASS 8/21 S=40000 (config 4)", 8, 21, 512, 40000, 0.1, 0.1, 1.0e8),
    ("YPMSD 20/90 S=40000 (config 5)", 20, 90, 512, 40000, 0.1, 0.1, 1.0e8),
    ("YPMSD 20/90 S=160000 (config 5)", 20, 90, 512, 160000, 0.1, 0.1, 1.0e8),
    ("YPMSD 20/90 S=640000 (config 5)", 20, 90, 512, 640000, 0.1, 0.1, 1.0e8),
    ("width sweep 8/21 g_units=64x10 (one-wave workgroups)", 8, 21, 64, 40000, 0.1, 0.1, 8.0e8),
    ("width sweep 8/21 g_units=64x10 RT_IS_NARROW=0 (four-wave kernel, padded to 128)", 8, 21, 64, 40000, 0.1, 0.1, 4.0e8),
    ("width sweep 8/21 g_units=128x10 (one-wave workgroups)", 8, 21, 128, 40000, 0.1, 0.1, 4.0e8),
    ("width sweep 8/21 g_units=128x10 RT_IS_NARROW=0 (four-wave kernel)", 8, 21, 128, 40000, 0.1, 0.1, 4.0e8),
    ("width sweep 2/2 g_units=64x10 (one-wave workgroups)", 2, 2, 64, 40000, 0.1, 0.5, 8.0e8),
    ("width sweep 2/2 g_units=128x10 (one-wave workgroups)", 2, 2, 128, 40000, 0.1, 0.5, 4.0e8),
    ("width sweep 8/21 g_units=256x10 (two-wave workgroups)", 8, 21, 256, 40000, 0.1, 0.1, 2.0e8),
    ("width sweep 8/21 g_units=256x10 RT_IS_NARROW=0 (four-wave kernel)", 8, 21, 256, 40000, 0.1, 0.1, 2.0e8),
    ("width sweep 2/2 g_units=256x10 (two-wave workgroups)", 2, 2, 256, 40000, 0.1, 0.5, 2.0e8),
    ("width sweep 8/21 g_units=512x10 (reference width)", 8, 21, 512, 40000, 0.1, 0.1, 1.0e8),
    ("width sweep 8/21 g_units=1024x10 (32-sample passes)", 8, 21, 1024, 40000, 0.1, 0.1, 0.25e8),
]

def main():
    eng = _lib.Engine(0)
    rows = []
    only = sys.argv[1] if len(sys.argv) > 1 else ""        # optional substring filter on the config name
    for name, dx, dy, hid, S, sd, scale, target in CONFIGS:
        if only not in name:
            continue
        os.environ["RT_IS_NARROW"] = "0" if "RT_IS_NARROW=0" in name else "1"      # read by rt_set_mlp
        rng = np.random.RandomState(1234)
        g = Generator(dx, dy, "g_net", nb_layers=10, nb_units=hid, rng=rng)
        h = Generator(dy, dx, "h_net", nb_layers=10, nb_units=256, rng=rng)
        eng.set_mlp(_lib.RT_MLP_G, g.weights, g.biases, g.alpha)
        eng.set_mlp(_lib.RT_MLP_H, h.weights, h.biases, h.alpha)
        N = max(256, int(target // S))
        y = np.random.RandomState(42).standard_normal((N, dy))
        y_dev, y32 = eng.to_device(y), eng.to_device(y.astype(np.float32))
        xh, out = eng.dev_alloc(N * dx * 4), eng.to_device(np.zeros(N))
        def step():
            eng.mlp_forward_dev(_lib.RT_MLP_H, y32, N, xh)
            eng.is_estimate_dev(y_dev, xh, N, S, sd, scale, out, df=1.0, seed=0, log=True)
        step(); eng.sync(); eng.reset_counters()
        t0 = time.perf_counter()
        for _ in range(2): step()
        eng.sync()
        dt = time.perf_counter() - t0
        c = eng.counters()
        tf = c["flops"] / (c["kernel_ms"] * 1e-3) / 1e12
        r = eng.from_device(out, (N,), np.float64)
        for ptr in (y_dev, y32, xh, out):
            eng.dev_free(ptr)
        rows.append({"config": name, "dx": dx, "dy": dy, "g_hidden": hid, "N": N, "S": S,
                     "flops_per_sample_eval": c["flops"] / c["sample_evals"],
                     "sample_evals_per_s": 2.0 * N * S / dt, "kernel_ms_avg": c["kernel_ms"] / c["launches"],
                     "tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK, "finite": bool(np.all(np.isfinite(r)))})
        print(json.dumps(rows[-1]), flush=True)

if __name__ == "__main__":
    main()
