"""GPU tests of the drop-in model objects (the reference's Python method signatures) against the oracle."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from roundtrip_amd import Generator, Roundtrip, RoundtripModel, _lib

pytestmark = pytest.mark.gpu


def _oracle_estimate(model, g, y, centres, S, sd, scale, df, log):
    xs = model.engine.draw_samples(centres, S, scale, df, seed=model.is_seed).astype(np.float64)
    og = c_oracle.OracleMlp(g.weights, g.biases, g.alpha)
    return c_oracle.is_estimate_given_samples(og, y, centres, xs, sd, scale, df, log, c_oracle.ORDER_KERNEL, 8)


def test_v2_estimate_px_with_IS(tmp_path):
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    params = dict(dataset="indep_gmm", output_dir=str(tmp_path), x_dim=2, z_dim=2, g_units=[512] * 10,
                  e_units=[256] * 10, dz_units=[128, 128], dx_units=[256] * 4, save_model=False, sd_x=0.05,
                  scale=0.5, sample_size=20000, lr=2e-4, alpha=10, beta=1, gamma=0, g_d_freq=1)
    m = Roundtrip(params, timestamp="gpu", random_seed=123)
    x = np.random.RandomState(0).standard_normal((9, 2)) * 0.2
    S = 3000
    got = m.estimate_px_with_IS(x, sd_x=params["sd_x"], scale=params["scale"], sample_size=S)
    assert got.shape == (9,) and got.dtype == np.float64
    z = m.e_net(x)                                            # self.e_net(x_points), roundtrip.py:283
    assert z.shape == (9, 2) and z.dtype == np.float32
    want = _oracle_estimate(m, m.g_net, x, z, S, 0.05, 0.5, 1.0, True)
    np.testing.assert_allclose(got, want, rtol=1e-12)
    lin = m.estimate_px_with_IS(x, sd_x=0.05, scale=0.5, sample_size=S, log=False)
    np.testing.assert_allclose(np.log(lin), got, rtol=1e-12)
    # same call => same answer (the reference reseeds with 0 at every call, roundtrip.py:230)
    assert np.array_equal(got, m.estimate_px_with_IS(x, sd_x=0.05, scale=0.5, sample_size=S, bs=7))
    # new weights are picked up
    m.g_net.set_weights([w * 0.5 for w in m.g_net.weights], m.g_net.biases)
    again = m.estimate_px_with_IS(x, sd_x=0.05, scale=0.5, sample_size=S)
    assert not np.array_equal(again, got)
    np.testing.assert_allclose(again, _oracle_estimate(m, m.g_net, x, z, S, 0.05, 0.5, 1.0, True), rtol=1e-12)
    # v1 spelling on the v2 object, with the save side effect (main_density_est.py:346-347)
    py = m.estimate_py_with_IS(x, 3, sd_y=0.05, scale=0.5, sample_size=S)
    assert np.array_equal(py, again)
    assert np.array_equal(np.load("%s/py_est_at_epoch3.npy" % m.save_dir), py)


def test_v1_model(tmp_path, monkeypatch):
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(1234)
    g = Generator(3, 6, "g_net", nb_layers=10, nb_units=512, rng=rng)      # main_density_est.py:447-448
    h = Generator(6, 3, "h_net", nb_layers=10, nb_units=256, rng=rng)
    m = RoundtripModel(g, h, None, None, None, None, "arem", None, 64, 10.0, 10.0, 3, False)
    y = np.random.RandomState(42).standard_normal((5, 6)) * 0.05
    xh = m.predict_x(y)                                                     # :237-249
    assert xh.dtype == np.float64 and xh.shape == (5, 3)
    oh = c_oracle.OracleMlp(h.weights, h.biases)
    assert np.array_equal(xh.astype(np.float32), oh.forward(y.astype(np.float32), c_oracle.ORDER_KERNEL))
    yy = m.predict_y(xh)
    og = c_oracle.OracleMlp(g.weights, g.biases)
    assert np.array_equal(yy.astype(np.float32), og.forward(xh.astype(np.float32), c_oracle.ORDER_KERNEL))
    S = 2000
    got = m.estimate_py_with_IS(y, 5, sd_y=0.1, scale=0.1, sample_size=S, save=True)     # df = self.df = 3
    want = _oracle_estimate(m, g, y, xh.astype(np.float32), S, 0.1, 0.1, 3.0, True)
    np.testing.assert_allclose(got, want, rtol=1e-12)
    assert np.array_equal(np.load("%s/py_est_at_epoch5.npy" % m.save_dir), got)
    assert len(os.listdir(m.save_dir)) == 1
    m.estimate_py_with_IS(y, 6, sample_size=64, save=False)
    assert len(os.listdir(m.save_dir)) == 1


def test_v1_model_selection_and_likelihood_summary(tmp_path, monkeypatch):
    """model_selection (main_density_est.py:207-219) and the validation bookkeeping (:186-189)."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(7)
    g = Generator(2, 2, "g_net", nb_layers=3, nb_units=128, rng=rng)
    h = Generator(2, 2, "h_net", nb_layers=3, nb_units=128, rng=rng)

    class Sampler(object):
        X_val = np.random.RandomState(8).standard_normal((40, 2)) * 0.3

    m = RoundtripModel(g, h, None, None, None, Sampler(), "toy", None, 64, 10.0, 10.0, 1, False)
    S = 500
    best_sd, best_scale = m.model_selection(sample_size=S)
    # the reference's way: 15 separate estimates
    records = []
    for sd in [0.05, 0.1, 0.5]:
        for scale in [0.005, 0.01, 0.1, 0.5, 1]:
            est = m.estimate_py_with_IS(Sampler.X_val, 0, sd_y=sd, scale=scale, sample_size=S, log=True, save=False)
            records.append([sd, scale, np.mean(est)])
    records.sort(key=lambda item: item[-1])
    assert (best_sd, best_scale) == (records[-1][0], records[-1][1])
    with open("log_val.txt", "a+") as f:
        mean, se = m.likelihood_summary(Sampler.X_val, 4, best_sd, best_scale, sample_size=S, log_file=f)
    est = m.estimate_py_with_IS(Sampler.X_val, 4, sd_y=best_sd, scale=best_scale, sample_size=S, save=False)
    assert mean == np.mean(est) and se == np.std(est) / np.sqrt(len(est))
    assert open("log_val.txt").read() == "%d\t%f\t%f\n" % (4, mean, se)


def _cf_from_jacobian(y, x_, y__, J, sd, log):
    """The reference's closed form (main_density_est.py:365-384), point by point, from given network outputs."""
    dx, dy = x_.shape[1], y.shape[1]
    out = []
    for n in range(len(y)):
        r = y[n] - y__[n]
        A = J[n].T @ J[n]
        b = J[n].T @ r
        Si = np.eye(dx) + A / sd ** 2
        Sg = np.linalg.inv(Si)
        mu = Sg.dot(b / sd ** 2 - x_[n])
        c = np.sum(x_[n] ** 2) + np.sum(r ** 2) / sd ** 2 - mu.T.dot(Si).dot(mu)
        if log:
            out.append(-dy * np.log(np.sqrt(2 * np.pi) * sd) + 0.5 * np.log(np.linalg.det(Sg)) - 0.5 * c)
        else:
            out.append(1. / (np.sqrt(2 * np.pi) * sd) ** dy * sd ** dy * np.sqrt(np.linalg.det(Sg)) * np.exp(-0.5 * c))
    return np.array(out)


def test_v1_estimate_py_with_CF(tmp_path, monkeypatch):
    """estimate_py_with_CF / get_jacobian (main_density_est.py:252-263, 351-387) on the v1 object."""
    from oracle import is_ref
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(1234)
    g = Generator(3, 6, "g_net", nb_layers=10, nb_units=512, rng=rng)
    h = Generator(6, 3, "h_net", nb_layers=10, nb_units=256, rng=rng)
    m = RoundtripModel(g, h, None, None, None, None, "arem", None, 64, 10.0, 10.0, 1, False)
    y = np.random.RandomState(42).standard_normal((37, 6)) * 0.3
    sd = 0.5
    got = m.estimate_py_with_CF(y, 2, sd_y=sd, log=True, save=True)
    assert got.shape == (37,) and got.dtype == np.float64
    assert np.array_equal(np.load("%s/py_est_at_epoch2.npy" % m.save_dir), got)
    # same network outputs (kernel-order oracle) through the reference's formulas, point by point
    og, oh = c_oracle.OracleMlp(g.weights, g.biases), c_oracle.OracleMlp(h.weights, h.biases)
    x_ = oh.forward(y.astype(np.float32), c_oracle.ORDER_KERNEL).astype(np.float64)
    y__, J = og.jacobian(x_, c_oracle.ORDER_KERNEL, nthreads=8)
    np.testing.assert_allclose(got, _cf_from_jacobian(y, x_, y__.astype(np.float64), J.astype(np.float64), sd, True),
                               rtol=1e-10)
    jac = m.get_jacobian(x_)
    assert jac.shape == (37, 6, 3) and np.array_equal(jac, J.astype(np.float64))
    # the numpy restatement of the whole estimator (natural summation order): north_star tolerance 1e-4
    want = is_ref.estimate_py_with_CF(is_ref.DenseNet(g.weights, g.biases), is_ref.DenseNet(h.weights, h.biases),
                                      y, sd_y=sd, log=True)
    np.testing.assert_allclose(got, want, rtol=1e-4)
    lin = m.estimate_py_with_CF(y, 2, sd_y=sd, log=False, save=False)
    np.testing.assert_allclose(np.log(lin) - 6 * np.log(sd), got, rtol=1e-9)       # the reference's extra sd^dy
    assert np.array_equal(m.estimate_px_with_CF(y, sd_x=sd), got)
    assert m.estimate_py_with_CF(np.zeros((0, 6)), 0, save=False).shape == (0,)


@pytest.mark.parametrize("name", ["indep_gmm_2_2", "arem_3_6", "ypmsd_20_90", "df3_biased_2_2"])
def test_cf_golden_vectors(name, tmp_path):
    from test_oracle import GOLDEN, golden_nets
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    gw, gb, hw, hb = golden_nets(f)
    dx, dy = int(f["dx"]), int(f["dy"])
    params = dict(dataset=name, output_dir=str(tmp_path), x_dim=dy, z_dim=dx, g_units=[512] * 10, e_units=[256] * 10,
                  save_model=False)
    m = Roundtrip(params, timestamp="cf")
    m.g_net.set_weights(gw, gb)
    m.e_net.set_weights(hw, hb)
    got = m.estimate_px_with_CF(f["y"], sd_x=float(f["sd"]), log=True)
    np.testing.assert_allclose(got, f["cf_log_est"], rtol=1e-4)
    jac = m.get_jacobian(f["xhat"])
    assert jac.shape == (len(f["y"]), dy, dx) and jac.dtype == np.float32
    assert np.abs(jac - f["jac_at_xhat"]).max() <= 1e-5 * np.abs(f["jac_at_xhat"]).max()


def test_reference_draws_mode_reproduces_the_single_threaded_reference(tmp_path, monkeypatch):
    """is_sampler = "numpy": numpy's legacy stream seeded with 0, consumed like sample_from_qx (main_density_est.py:267,
    310-318), scored on the GPU.  The oracle's statement-by-statement restatement of estimate_py_with_IS draws from the
    same stream (the GLOBAL generator, as the reference does) and runs G in numpy: the two must agree to the fp32
    rounding of G, for both classes, log and linear, and for a rank that starts in the middle of the stream."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    from oracle import is_ref
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(5)
    g = Generator(2, 2, "g_net", nb_layers=10, nb_units=512, rng=rng)
    h = Generator(2, 2, "h_net", nb_layers=10, nb_units=256, rng=rng)
    m = RoundtripModel(g, h, data="indep_gmm", df=1)
    m.is_sampler = "numpy"
    y = np.random.RandomState(6).standard_normal((7, 2)) * 0.3
    S = 2500
    state = np.random.get_state()
    got = m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.5, sample_size=S, log=True, save=False)
    assert np.array_equal(state[1], np.random.get_state()[1]) and state[2] == np.random.get_state()[2]   # global generator untouched
    # the oracle's own numpy networks: a product Generator handed to the oracle would run its forward on the GPU
    og, oh = is_ref.DenseNet(g.weights, g.biases, g.alpha), is_ref.DenseNet(h.weights, h.biases, h.alpha)
    want = is_ref.estimate_py_with_IS(og, oh, y, sd_y=0.1, scale=0.5, sample_size=S, df=1, log=True)
    np.testing.assert_allclose(got, want, rtol=2e-6)
    lin = m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.5, sample_size=S, log=False, save=False)
    np.testing.assert_allclose(lin, is_ref.estimate_py_with_IS(og, oh, y, sd_y=0.1, scale=0.5, sample_size=S, df=1,
                                                               log=False), rtol=2e-4)
    # a block that starts at global index 3 consumes the first three points' draws first
    blk = m._reference_draws_block(y, 0.1, 0.5, S, 1, True)
    np.testing.assert_array_equal(blk(y[3:], 3), got[3:])
    # df = 3 (chisquare with another shape), through the v2 class
    params = dict(dataset="t", output_dir=str(tmp_path), x_dim=2, z_dim=2, g_units=[512] * 10, e_units=[256] * 10)
    m2 = Roundtrip(params, timestamp="ref", random_seed=9)
    m2.is_sampler = "numpy"
    got3 = m2.estimate_px_with_IS(y, sd_x=0.2, scale=0.3, sample_size=1500, df=3)
    want3 = is_ref.estimate_py_with_IS(is_ref.DenseNet(m2.g_net.weights, m2.g_net.biases, m2.g_net.alpha),
                                       is_ref.DenseNet(m2.e_net.weights, m2.e_net.biases, m2.e_net.alpha), y, sd_y=0.2,
                                       scale=0.3, sample_size=1500, df=3)
    np.testing.assert_allclose(got3, want3, rtol=2e-6)
    m2.is_sampler = "mt"
    with pytest.raises(ValueError):
        m2.estimate_px_with_IS(y)


def test_model_level_sharding_two_ranks(tmp_path):
    """One process per rank (here: two gloo ranks that share HIP device 0; on a node it is one GPU each over RCCL):
    the drop-in classes shard the query points, every rank returns the FULL result, and it equals the one-process
    answer bit for bit (contiguous blocks keyed by the global query index) — IS, the several-sd sweep and the
    Laplace estimator."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    import socket, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        os.environ["LOCAL_RANK_SAVED"] = os.environ.get("LOCAL_RANK", "0")
        import numpy as np
        import torch.distributed as dist
        from roundtrip_amd import Generator, RoundtripModel
        dist.init_process_group("gloo")
        rng = np.random.RandomState(21)
        g = Generator(3, 6, "g_net", nb_layers=10, nb_units=512, rng=rng)
        h = Generator(6, 3, "h_net", nb_layers=10, nb_units=256, rng=rng)
        m = RoundtripModel(g, h, data="arem", df=1, device=0)        # both ranks on the one GPU of the test box
        y = np.random.RandomState(22).standard_normal((13, 6)) * 0.1
        out = dict(IS=m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False),
                   multi=m._estimate_is_multi(y, [0.05, 0.1], 0.1, 900, 1, True),
                   CF=m.estimate_py_with_CF(y, 0, sd_y=0.1, save=False))
        m.shard = False
        one = dict(IS=m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False),
                   multi=m._estimate_is_multi(y, [0.05, 0.1], 0.1, 900, 1, True),
                   CF=m.estimate_py_with_CF(y, 0, sd_y=0.1, save=False))
        for k in out:
            assert out[k].shape == one[k].shape and np.array_equal(out[k], one[k]), (k, out[k], one[k])
        # fewer query points than ranks: the SAMPLES of the point are split over the ranks (automatic), same bits
        m.shard = True
        for S in (900, 650):
            split = m.estimate_py_with_IS(y[:1], 0, sd_y=0.1, scale=0.1, sample_size=S, save=False)
            m.shard = False
            alone = m.estimate_py_with_IS(y[:1], 0, sd_y=0.1, scale=0.1, sample_size=S, save=False)
            m.shard = True
            assert np.array_equal(split, alone), (S, split, alone)
        m.split_samples = True                                        # forced, with more points than ranks
        assert np.array_equal(m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False), one["IS"])
        dist.barrier(); dist.destroy_process_group()
        print("rank ok")
    """ % (root,)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300,
                         cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count("rank ok") == 2


def _run_script(tmp_path, body, timeout=300, env=None):
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text("import os, sys\nsys.path.insert(0, %r)\n" % root + textwrap.dedent(body))
    return subprocess.run([sys.executable, str(script)], env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})),
                          capture_output=True, text=True, timeout=timeout, cwd=str(tmp_path))


def test_rccl_branch_world_size_1(tmp_path):
    """The real RCCL path (rt_comm_* behind the C ABI, librccl dlopen'ed, the engine's device and stream) with a
    one-rank communicator — all a one-GPU box allows, RCCL refusing two ranks on one GPU: dist.init from a launcher's
    environment, all_gather_blocks, all_gather_ranges, the device-resident gather, the max-reduce and the barrier, and
    the drop-in class's sharded estimate_py_with_IS / split-samples / several-sd / Laplace paths through the gather.
    Runs in a process of its own, which never imports torch."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    res = _run_script(tmp_path, """
        import numpy as np
        from roundtrip_amd import Generator, RoundtripModel, _lib, dist as rdist
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
        rng = np.random.RandomState(21)
        g = Generator(3, 6, "g_net", nb_layers=10, nb_units=512, rng=rng)
        h = Generator(6, 3, "h_net", nb_layers=10, nb_units=256, rng=rng)
        m = RoundtripModel(g, h, data="arem", df=1, device=0)
        y = np.random.RandomState(22).standard_normal((13, 6)) * 0.1
        one = dict(IS=m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False),
                   multi=m._estimate_is_multi(y, [0.05, 0.1], 0.1, 900, 1, True),
                   CF=m.estimate_py_with_CF(y, 0, sd_y=0.1, save=False))
        assert not rdist.active() and rdist.world() == (0, 1)
        assert m.init_distributed() == (0, 1)
        assert rdist.active() and m.engine.comm_info() == (0, 1)
        eng = m.engine
        # the raw collectives
        v = np.arange(7, dtype=np.float64) * 1.5
        assert np.array_equal(eng.comm_all_gather(v), v[None, :])
        assert np.array_equal(eng.comm_all_reduce_max([3.25, -1.0]), [3.25, -1.0])
        eng.comm_barrier()
        src, dst = eng.to_device(v), eng.dev_alloc(v.nbytes)
        eng.comm_all_gather_dev(src, v.size, dst)
        eng.sync()
        assert np.array_equal(eng.from_device(dst, v.shape, np.float64), v)
        eng.dev_free(src); eng.dev_free(dst)
        # dist's gathers
        assert np.array_equal(rdist.all_gather_blocks(v, 7), v)
        whole = np.random.RandomState(1).standard_normal((3, 5, 2, 2))
        assert np.array_equal(rdist.all_gather_ranges(whole, 5, axis=1), whole)
        calls = []
        out = rdist.sharded_map(y, lambda blk, base: (calls.append((base, len(blk))), blk.sum(axis=1))[1])
        assert calls == [(0, 13)] and np.array_equal(out, y.sum(axis=1))
        # the drop-in class through the gather
        got = dict(IS=m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False),
                   multi=m._estimate_is_multi(y, [0.05, 0.1], 0.1, 900, 1, True),
                   CF=m.estimate_py_with_CF(y, 0, sd_y=0.1, save=False))
        for k in got:
            assert got[k].shape == one[k].shape and np.array_equal(got[k], one[k]), k
        m.split_samples = True
        assert np.array_equal(m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False), one["IS"])
        try:
            rdist.init(eng)
            raise SystemExit("second init must be refused")
        except RuntimeError:
            pass
        rdist.shutdown()
        assert not rdist.active()
        assert "torch" not in sys.modules
        print("rccl ok")
    """)
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment starts two rank processes itself and prints ONE
    JSON line with n_gpus = 2 (gloo on host tensors and one shared GPU here: the rehearsal mode; on a node it is one GPU
    per rank over RCCL); `--gpus 1` prints the line with the roofline and both CPU legs."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--steps", "1", "--warmup", "0", "--grid", "16", "--sample-size", "640"],
                         env=dict(env, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["query_points_total"] == 512 and d["result_check"]["finite"]
    assert d["value"] > 0 and d["roofline"]["launches"] == 1
    res = subprocess.run([sys.executable, "-X", "importtime", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1",
                          "--warmup", "0", "--grid", "16", "--sample-size", "640", "--no-live-traffic", "--cpu-points", "8",
                          "--cpu-points-numpy", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    imported = [l.split("|")[-1].strip() for l in res.stderr.splitlines() if l.startswith("import time:")]
    assert "numpy" in imported and "roundtrip_amd" in imported and not [m for m in imported if m.split(".")[0] == "torch"]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0
    assert "matches GPU on this sample: True" in d["cpu_baseline"]["sample"]
    assert "matches GPU on this sample: True" in d["cpu_baseline"]["numpy_restatement"]["sample"]


_COMM_DOUBLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "comm_double", "librccl_double.so")


def test_two_ranks_through_rt_comm(tmp_path):
    """N = 2 through the SAME code the multi-GPU node runs — dist.init's rendezvous, rt_comm_init, the gathers sized
    per * world, the max-reduce, the barrier — on two ranks that share the one GPU of the test box.  RCCL refuses that, so
    RT_RCCL_LIB points rt_is.hip's loader at tests/comm_double (the six nccl* entry points over shared memory): this
    checks THIS repo's side of the collective (counts, offsets, block order, y_index_base) bit for bit against the
    one-process answers, not RCCL.  No torch in the rank processes."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    assert os.path.exists(_COMM_DOUBLE), "build() compiles tests/comm_double"
    import socket, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text("import os, sys\nsys.path.insert(0, %r)\n" % root + textwrap.dedent("""
        import numpy as np
        from roundtrip_amd import Generator, RoundtripModel, dist as rdist
        rank = int(os.environ["RANK"])
        rng = np.random.RandomState(21)
        g = Generator(3, 6, "g_net", nb_layers=10, nb_units=512, rng=rng)
        h = Generator(6, 3, "h_net", nb_layers=10, nb_units=256, rng=rng)
        m = RoundtripModel(g, h, data="arem", df=1, device=0)        # both ranks on the one GPU of the test box
        y = np.random.RandomState(22).standard_normal((13, 6)) * 0.1
        def run():
            return dict(IS=m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False),
                        multi=m._estimate_is_multi(y, [0.05, 0.1], 0.1, 900, 1, True),
                        CF=m.estimate_py_with_CF(y, 0, sd_y=0.1, save=False))
        one = run()                                                  # before init: one process, all 13 points
        alone = {S: m.estimate_py_with_IS(y[:1], 0, sd_y=0.1, scale=0.1, sample_size=S, save=False) for S in (900, 650)}
        assert m.init_distributed() == (rank, 2) and rdist.world() == (rank, 2) and m.engine.comm_info() == (rank, 2)
        eng = m.engine
        # the raw collectives: rows arrive in rank order
        v = np.arange(5, dtype=np.float64) + 10.0 * rank
        assert np.array_equal(eng.comm_all_gather(v), np.stack([np.arange(5.0), np.arange(5.0) + 10.0]))
        assert np.array_equal(eng.comm_all_reduce_max([float(rank), -float(rank)]), [1.0, 0.0])
        src, dst = eng.to_device(v), eng.dev_alloc(2 * v.nbytes)
        eng.comm_all_gather_dev(src, v.size, dst)
        eng.sync()
        assert np.array_equal(eng.from_device(dst, (2, 5), np.float64), np.stack([np.arange(5.0), np.arange(5.0) + 10.0]))
        eng.dev_free(src); eng.dev_free(dst)
        eng.comm_barrier()
        # ragged blocks: 13 points = 7 + 6, 3 points = 2 + 1, 1 point = 1 + 0
        for n in (13, 3, 1):
            lo, hi = rdist.shard_bounds(n, 2, rank)
            assert np.array_equal(rdist.all_gather_blocks(np.arange(lo, hi, dtype=np.float64), n), np.arange(n))
        whole = np.random.RandomState(1).standard_normal((3, 5, 2, 2))
        lo, hi = rdist.shard_bounds(5, 2, rank)
        assert np.array_equal(rdist.all_gather_ranges(whole[:, lo:hi], 5, axis=1), whole)
        # the drop-in class: sharded query points, every rank gets the full result, the one-process bits
        got = run()
        for k in got:
            assert got[k].shape == one[k].shape and np.array_equal(got[k], one[k]), (k, got[k], one[k])
        for S in (900, 650):                                         # fewer points than ranks: samples split
            assert np.array_equal(m.estimate_py_with_IS(y[:1], 0, sd_y=0.1, scale=0.1, sample_size=S, save=False), alone[S])
        # ranks of unequal speed: blocks cut 3 : 1 (10 + 3 points) and 1 : 3, a padded all-gather — the same bits
        for w in ([3.0, 1.0], [1.0, 3.0]):
            rdist.set_rank_weights(w)
            lo, hi = rdist.shard_bounds(13, 2, rank, rdist.rank_weights())
            assert (hi - lo) == ((10, 3) if w[0] > w[1] else (3, 10))[rank]
            got = run()
            for k in got:
                assert got[k].shape == one[k].shape and np.array_equal(got[k], one[k]), (w, k)
        rdist.set_rank_weights(None)
        m.split_samples = True
        assert np.array_equal(m.estimate_py_with_IS(y, 0, sd_y=0.1, scale=0.1, sample_size=900, save=False), one["IS"])
        rdist.shutdown()
        assert "torch" not in sys.modules
        print("rank %d ok" % rank)
    """))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RT_COMM_RUN_ID="t-%d" % os.getpid(), RT_RCCL_LIB=_COMM_DOUBLE, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=280))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in so, "rank %d: %s %s" % (r, so[-2000:], se[-3000:])


def test_bench_two_ranks_default_backend(tmp_path):
    """`python bench.py --gpus 2` on its default backend (rt_comm_*: device-pointer gather on the engine's stream, barrier
    and max-over-ranks through the C ABI), the two ranks sharing the box's one GPU and the communicator coming from
    tests/comm_double (see above).  One JSON line, n_gpus = 2, every result finite; and the gathered result is the
    two-block answer: its mean equals the mean of a one-GPU run over the same 2 x 256 points."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2",
                          "--warmup", "1", "--grid", "16", "--sample-size", "640"],
                         env=dict(env, RT_RCCL_LIB=_COMM_DOUBLE), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["query_points_total"] == 512 and d["result_check"]["finite"]
    assert d["config"]["collective"] == "rccl all-gather" and d["steps"] == 2 and d["value"] > 0
    # the driver's launch line for N > 1 (one rank per process under torch.distributed.run): the same line again
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--grid", "16", "--sample-size", "640"],
                         env=dict(env, RT_RCCL_LIB=_COMM_DOUBLE, OMP_NUM_THREADS="1"), capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["result_check"] == d["result_check"] and d2["config"] == d["config"]
    # the N > 1 line explains itself: every rank's kernel and wall time per step, the gather alone, the slowest rank
    for line in (d, d2):
        r = line["ranks"]
        assert len(r["per_rank_kernel_ms"]) == 2 and len(r["per_rank_step_ms"]) == 2 and r["per_rank_query_points"] == [256, 256]
        assert min(r["per_rank_kernel_ms"]) > 0 and r["slowest_rank"] in (0, 1) and r["gather_ms"] > 0
        assert 0.5 <= r["efficiency_vs_slowest"] <= 1.0 and r["rank_weights"] is None
        assert line["roofline"]["kernel"] == "is_fused_kernel<512, 2, false, false, 4>"
    # --balance-ranks: blocks re-cut by the rates measured on a warm-up step (two ranks contending for ONE GPU here, so the
    # rates are whatever they are); the gathered result does not change by a bit
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "1",
                          "--warmup", "1", "--grid", "16", "--sample-size", "640", "--balance-ranks"],
                         env=dict(env, RT_RCCL_LIB=_COMM_DOUBLE), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d3 = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d3["result_check"] == d["result_check"] and sum(d3["ranks"]["per_rank_query_points"]) == 512
    if d3["ranks"]["rank_weights"] is not None:
        assert len(d3["ranks"]["rank_weights"]) == 2
    # the same 512 global points (the 16 x 16 tile twice, global indices 0..511) evaluated by one process
    import numpy as np
    sys.path.insert(0, root)
    import bench as B
    from roundtrip_amd import _lib
    dx, dy, sd, scale, _, log = B.WORKLOADS["grid"][:6]
    g, h = B.build_nets(dx, dy)
    eng = _lib.Engine(0)
    eng.set_mlp(_lib.RT_MLP_G, g.weights, g.biases, g.alpha)
    eng.set_mlp(_lib.RT_MLP_H, h.weights, h.biases, h.alpha)
    y = np.tile(B.grid_points(16), (2, 1))
    xhat = eng.mlp_forward(_lib.RT_MLP_H, y.astype(np.float32))
    want = eng.is_estimate(y, xhat, 640, sd, scale, df=1.0, seed=0, log=log)
    eng.close()
    assert d["result_check"]["mean"] == float(want.mean())


@pytest.mark.parametrize("args", [["--workload", "hepmass"], ["--workload", "bank"], ["--workload", "arem"],
                                  ["--workload", "involute", "--grid", "16"], ["--workload", "indep-gmm", "--grid", "16"],
                                  ["--workload", "hepmass", "--g-units", "64"], ["--workload", "hepmass", "--g-units", "128"],
                                  ["--workload", "hepmass", "--g-units", "256"], ["--workload", "hepmass", "--g-units", "1024"]])
def test_bench_workloads_smoke(args):
    """Every BASELINE row bench.py can run (--workload, --g-units) at 256 query points: one JSON line with the roofline
    (kernel name from the library) and the CPU legs, which agree with the GPU on their sample."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU on this machine")
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--points", "256",
                          "--sample-size", "640", "--no-live-traffic", "--cpu-points", "8", "--cpu-points-numpy", "2"] + args,
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["query_points_total"] == 256 and d["result_check"]["finite"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0 and d["roofline"]["launches"] == 1
    width = int(args[args.index("--g-units") + 1]) if "--g-units" in args else 512
    assert d["roofline"]["kernel"].startswith("is_fused_kernel<%d," % width)
    assert "matches GPU on this sample: True" in d["cpu_baseline"]["sample"]
    assert "matches GPU on this sample: True" in d["cpu_baseline"]["numpy_restatement"]["sample"]
