import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roundtrip_amd._lib import Engine, RT_MLP_G, RT_MLP_H
from oracle import is_ref, c_oracle

def main():
    dx, dy = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rng = np.random.RandomState(1234)
    gw, gb = is_ref.glorot_mlp([dx] + [512]*10 + [dy], rng)
    hw, hb = is_ref.glorot_mlp([dy] + [256]*10 + [dx], rng)
    gb = [np.random.RandomState(5).uniform(-.1,.1,size=b.shape).astype(np.float32) for b in gb]
    eng = Engine(0)
    print("CUs", eng.cu_count())
    eng.set_mlp(RT_MLP_G, gw, gb); eng.set_mlp(RT_MLP_H, hw, hb)
    og = c_oracle.OracleMlp(gw, gb); oh = c_oracle.OracleMlp(hw, hb)
    x = np.random.RandomState(3).standard_normal((200, dx)).astype(np.float32) * 2
    yg = eng.mlp_forward(RT_MLP_G, x)
    yo = og.forward(x, order=c_oracle.ORDER_KERNEL); yn = og.forward(x, order=0)
    print("G fwd: bitexact vs kernel-order oracle:", np.array_equal(yg, yo), "max abs diff", np.abs(yg-yo).max(), "vs natural", np.abs(yg-yn).max(), "scale", np.abs(yn).max())
    y = np.random.RandomState(4).standard_normal((70, dy))
    xh = eng.mlp_forward(RT_MLP_H, y.astype(np.float32))
    xo = oh.forward(y.astype(np.float32), order=1)
    print("H fwd bitexact:", np.array_equal(xh, xo), np.abs(xh-xo).max())
    # given samples
    S = 1000
    np.random.seed(0)
    xs = np.stack([is_ref.sample_from_qx(xh[n].astype(np.float64), dx, 0.5, 1, S) for n in range(8)])
    for log in (True, False):
        got = eng.is_estimate_given_samples(y[:8], xh[:8], xs, 0.5, 0.5, 1.0, log=log)
        ref = c_oracle.is_estimate_given_samples(og, y[:8], xh[:8], xs, 0.5, 0.5, 1.0, log=log, order=1)
        print("given log=%s" % log, got[:4], ref[:4], "max rel", np.max(np.abs(got-ref)/np.abs(ref)))
    # rng mode
    S = 4096
    got = eng.is_estimate(y[:8], xh[:8], S, 0.5, 0.5, 1.0, seed=0)
    xs32 = eng.draw_samples(xh[:8], S, 0.5, 1.0, seed=0)
    ref = c_oracle.is_estimate_given_samples(og, y[:8], xh[:8], xs32.astype(np.float64), 0.5, 0.5, 1.0, order=1, nthreads=8)
    print("rng mode", got[:4], ref[:4], "max rel", np.max(np.abs(got-ref)/np.abs(ref)))
    xs_o = c_oracle.draw_samples(xh[:8], 0, S, 0.5, 1.0, 0)
    print("sampler vs oracle max rel diff", np.max(np.abs(xs32-xs_o)/(np.abs(xs_o)+1e-6)))
    # timing
    for N, S in ((256, 4096), (1024, 40000)):
        yy = np.random.RandomState(7).standard_normal((N, dy)); xx = eng.mlp_forward(RT_MLP_H, yy.astype(np.float32))
        eng.reset_counters()
        t0 = time.time(); out = eng.is_estimate(yy, xx, S, 0.1, 0.1, 1.0); t1 = time.time()
        c = eng.counters()
        print("N=%d S=%d wall %.3fs kernel %.3f ms  %.2f Meval/s  %.1f TFLOP/s (%.1f%% of 157.3)" % (N, S, t1-t0, c["kernel_ms"], c["sample_evals"]/c["kernel_ms"]/1e3, c["flops"]/c["kernel_ms"]/1e9, c["flops"]/c["kernel_ms"]/1e9/157.3*100), "finite", np.isfinite(out).all())
main()
