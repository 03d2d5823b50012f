"""Drop-in model objects for the importance-sampling density estimator.

`Roundtrip` mirrors the v2 class (/root/reference/src/pyroundtrip/roundtrip.py:9-58, 229-310): same
constructor arguments, same `params` keys, same attributes on this path (g_net, e_net, params,
save_dir, checkpoint_path) and the same `estimate_px_with_IS` signature.  `RoundtripModel` mirrors the
v1 class (/root/reference/main_density_est.py:29-44, 222-249, 266-348, 397-411).  Both spellings of the
estimator exist on both classes (BASELINE.json's north_star mixes them).

Everything numerical runs in librt_is.so on an MI355X; there is no CPU path in this package.
"""
import datetime
import os

import numpy as np

from . import _lib, dist
from .nets import BaseFullyConnectedNet, DenseNet


def _default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


class _IsModelBase(object):
    """Shared machinery: the device engine, lazy weight upload, the estimator itself."""

    #: seed of the counter-based proposal sampler.  The reference reseeds numpy's global RNG with 0 at
    #: every call (main_density_est.py:267), i.e. "same call => same answer"; a fixed default keeps that.
    is_seed = 0
    #: shard query points over the ranks of the job (one process per GPU) when there are several: after
    #: init_distributed() (RCCL behind the C ABI, no torch), or under a torch.distributed process group the caller has
    #: initialised itself
    shard = True
    #: where the proposal draws come from.  "philox": the in-kernel counter-based sampler (the fast path).
    #: "numpy": the reference's own draws — numpy's legacy MT19937 stream seeded with 0, consumed exactly as
    #: sample_from_qx does (main_density_est.py:267, 310-318: per query point, in order, chisquare(df, S) then
    #: multivariate_normal(0, scale^2 I, (S,))) — generated on the host and scored by the same kernel
    #: (rt_is_estimate_given_samples).  With identical weights this reproduces a single-threaded run of the reference
    #: up to the fp32 rounding of G; it is ~100x slower than "philox" (the host RNG) and exists for validation.
    is_sampler = "philox"
    #: with fewer query points than ranks, split every point's SAMPLES over the ranks instead of leaving ranks idle: each
    #: rank computes a range of a point's per-pass (max, sum) pairs, one all-gather exchanges them, every rank merges them
    #: in the fixed order — bit-identical to the one-process answer.  None: automatic (N < world size); True / False force.
    split_samples = None

    def _init_engine_state(self, device):
        self._device = device
        self._engine = None
        self._uploaded = {}

    # ---- engine ---------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            dev = self._device if self._device is not None else _default_device()
            self._engine = _lib.Engine(dev)   # raises if librt_is.so or the GPU is missing
        return self._engine

    def init_distributed(self, rank=None, world_size=None, addr=None, port=None):
        """Collective over all ranks of the job: makes this model's engine a member of the job's RCCL communicator
        (arguments default to the launcher's RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  The estimators then shard
        query points over the ranks and every rank returns the full result."""
        return dist.init(self.engine, rank, world_size, addr, port)

    def _nets(self):
        raise NotImplementedError

    def _sync_weights(self):
        eng = self.engine
        for which, net in self._nets():
            if self._uploaded.get(which) != (id(net), net.version):
                eng.set_mlp(which, net.weights, net.biases, net.alpha)
                self._uploaded[which] = (id(net), net.version)

    def _bind(self):
        for which, net in self._nets():
            net._forward = (lambda x, _w=which: self._forward(_w, x))

    def _forward(self, which, x):
        self._sync_weights()
        return self.engine.mlp_forward(which, x)

    # ---- the estimator ----------------------------------------------------------------------------
    def _estimate_is(self, points, sd, scale, sample_size, df, log):
        g_net = self._nets()[0][1]
        e_net = self._nets()[1][1]
        points = np.asarray(points)
        assert points.ndim == 2 and points.shape[-1] == g_net.output_dim
        points = np.ascontiguousarray(points, dtype=np.float64)
        self._sync_weights()
        eng = self.engine

        def block(pts, base):
            # proposal centres H(y): predict_x (main_density_est.py:320) / self.e_net(x_points) (roundtrip.py:283)
            centres = eng.mlp_forward(_lib.RT_MLP_H, pts.astype(np.float32))
            return eng.is_estimate(pts, centres, sample_size, sd, scale, df=df, seed=self.is_seed, log=log,
                                   y_index_base=base)

        if self.is_sampler == "numpy":
            block = self._reference_draws_block(points, sd, scale, sample_size, df, log)
        elif self.is_sampler != "philox":
            raise ValueError("is_sampler must be 'philox' or 'numpy', not %r" % (self.is_sampler,))
        ws = dist.world()[1]
        if self.shard and dist.active() and self.is_sampler == "philox" and points.shape[0] > 0 and \
                (self.split_samples if self.split_samples is not None else points.shape[0] < ws):
            return self._estimate_is_split(points, [sd], scale, sample_size, df, log)[0]
        if self.shard and dist.active():
            return dist.sharded_map(points, block, device=eng.device)
        if points.shape[0] == 0:
            return np.zeros(0, dtype=np.float64)
        return block(points, 0)

    def _estimate_is_split(self, points, sds, scale, sample_size, df, log):
        """(len(sds), N): every rank evaluates ALL query points on its range of each point's sample passes."""
        eng = self.engine
        rank, ws = dist.world()
        centres = eng.mlp_forward(_lib.RT_MLP_H, points.astype(np.float32))
        E = eng.is_pair_count(sample_size)
        lo, hi = dist.shard_bounds(E, ws, rank)
        mine = eng.is_pairs(points, centres, sample_size, sds, scale, lo, hi, df=df, seed=self.is_seed)
        pairs = dist.all_gather_ranges(mine, E, axis=1, device=eng.device)
        return eng.is_merge_pairs(pairs, sample_size, log=log)

    def _reference_draws_block(self, points, sd, scale, sample_size, df, log):
        """block(pts, base) for is_sampler == "numpy".  The legacy stream is sequential over ALL query points, so a
        rank that starts at `base` first consumes (and discards) the draws of the points before it."""
        eng = self.engine
        x_dim = self._nets()[0][1].input_dim
        S = int(sample_size)
        cov = np.diag(scale ** 2 * np.ones(x_dim))
        bs = max(1, min(1024, (256 << 20) // (S * x_dim * 8)))   # <= 256 MB of float64 draws per upload

        def draw(rng, centre):
            # main_density_est.py:315-318 / roundtrip.py:278-280, same calls in the same order
            z1 = rng.chisquare(df, S) / df
            z2 = rng.multivariate_normal(np.zeros(x_dim), cov, (S,))
            return centre + z2 / np.sqrt(z1)[:, None]

        def block(pts, base):
            rng = np.random.RandomState(0)          # np.random.seed(0) at :267, without clobbering the global generator
            centres = eng.mlp_forward(_lib.RT_MLP_H, pts.astype(np.float32))
            for _ in range(base):
                draw(rng, 0.0)
            out = np.empty(pts.shape[0], dtype=np.float64)
            for lo in range(0, pts.shape[0], bs):
                hi = min(pts.shape[0], lo + bs)
                xs = np.stack([draw(rng, centres[i].astype(np.float64)) for i in range(lo, hi)])
                out[lo:hi] = eng.is_estimate_given_samples(pts[lo:hi], centres[lo:hi], xs, sd, scale, df, log=log)
            return out
        return block

    # ---- the Laplace (closed-form) estimator ----------------------------------------------------------
    def _jacobian(self, latent):
        """(G(x) (N, y_dim), J (N, y_dim, x_dim)) as float64 containers of float32 values, like predict_y
        (main_density_est.py:225) and get_jacobian (:254).  Forward-mode differentiation on the GPU."""
        self._sync_weights()
        out, J = self.engine.mlp_jacobian(_lib.RT_MLP_G, np.ascontiguousarray(latent, dtype=np.float32))
        return out.astype(np.float64), J.astype(np.float64)

    def _estimate_cf(self, points, sd, log):
        """main_density_est.py:351-387 (v2: roundtrip.py:327-369), everything on the GPU (rt_cf_estimate): H(y), G(H(y)),
        the Jacobian of G at H(y), and the float64 x_dim x x_dim algebra per query point; the Jacobians never leave the
        device."""
        g_net = self._nets()[0][1]
        points = np.asarray(points)
        assert points.ndim == 2 and points.shape[-1] == g_net.output_dim
        points = np.ascontiguousarray(points, dtype=np.float64)
        self._sync_weights()
        eng = self.engine

        def block(pts, base):
            return eng.cf_estimate(pts, sd, log=log)

        if self.shard and dist.active():
            return dist.sharded_map(points, block, device=eng.device)
        if points.shape[0] == 0:
            return np.zeros(0, dtype=np.float64)
        return block(points, 0)

    def _estimate_is_multi(self, points, sds, scale, sample_size, df, log):
        """(len(sds), N): the estimate for several sd values from ONE sampler + G pass (same draws, as in the
        reference, which reseeds with 0 before every call)."""
        g_net = self._nets()[0][1]
        points = np.asarray(points)
        assert points.ndim == 2 and points.shape[-1] == g_net.output_dim
        points = np.ascontiguousarray(points, dtype=np.float64)
        sds = [float(v) for v in sds]
        self._sync_weights()
        eng = self.engine
        if points.shape[0] == 0:
            return np.zeros((len(sds), 0), dtype=np.float64)
        rank, ws = dist.world()
        if not (self.shard and dist.active()):
            centres = eng.mlp_forward(_lib.RT_MLP_H, points.astype(np.float32))
            return eng.is_estimate_multi(points, centres, sample_size, sds, scale, df=df, seed=self.is_seed, log=log)
        lo, hi = dist.shard_bounds(points.shape[0], ws, rank, dist.rank_weights())
        local = np.zeros((len(sds), 0))
        if hi > lo:
            centres = eng.mlp_forward(_lib.RT_MLP_H, points[lo:hi].astype(np.float32))
            local = eng.is_estimate_multi(points[lo:hi], centres, sample_size, sds, scale, df=df,
                                          seed=self.is_seed, log=log, y_index_base=lo)
        return np.stack([dist.all_gather_blocks(local[k], points.shape[0], device=eng.device) for k in range(len(sds))])


class Roundtrip(_IsModelBase):
    """Roundtrip model for density estimation — the importance-sampling path (v2 API)."""

    def __init__(self, params, timestamp=None, random_seed=None, device=None):
        super(Roundtrip, self).__init__()
        self.params = params
        self.timestamp = timestamp
        rng = np.random.RandomState(random_seed)
        self.g_net = BaseFullyConnectedNet(input_dim=params['z_dim'], output_dim=params['x_dim'],
                                           model_name='g_net', nb_units=params['g_units'], rng=rng)
        self.e_net = BaseFullyConnectedNet(input_dim=params['x_dim'], output_dim=params['z_dim'],
                                           model_name='e_net', nb_units=params['e_units'], rng=rng)
        # dz_net / dx_net / optimizers (roundtrip.py:22-29) belong to training, which is out of scope.
        if self.timestamp is None:
            now = datetime.datetime.now().astimezone()
            self.timestamp = now.strftime('%Y%m%d_%H%M%S')
        out_dir = params.get('output_dir', '.')
        dataset = params.get('dataset', 'data')
        self.checkpoint_path = "{}/checkpoints/{}/{}".format(out_dir, dataset, self.timestamp)
        if params.get('save_model', False) and not os.path.exists(self.checkpoint_path):
            os.makedirs(self.checkpoint_path)
        self.save_dir = "{}/results/{}/{}".format(out_dir, dataset, self.timestamp)
        if not os.path.exists(self.save_dir):
            os.makedirs(self.save_dir)
        self._init_engine_state(device)
        self._bind()
        # roundtrip.py:56-58: restore the latest checkpoint under checkpoint_path if there is one
        from . import checkpoint
        latest = checkpoint.latest_checkpoint(self.checkpoint_path)
        if latest:
            self.restore(latest)
            print('Latest checkpoint restored!!')

    def get_config(self):
        return {"params": self.params}

    def _nets(self):
        return ((_lib.RT_MLP_G, self.g_net), (_lib.RT_MLP_H, self.e_net))

    def restore(self, prefix):
        """Loads g_net / e_net from a tf.train.Checkpoint written by the reference (roundtrip.py:48-57)."""
        from . import checkpoint
        checkpoint.load_v2(prefix, self.g_net, self.e_net)

    def estimate_px_with_IS(self, x_points, sd_x=0.5, scale=0.5, sample_size=40000, bs=1024, df=1, log=True):
        """p(x) (or log p(x)) of every row of x_points by importance sampling (roundtrip.py:229-310).
        `bs` only batches work in the reference and has no effect here."""
        x_points = np.asarray(x_points)
        assert x_points.shape[-1] == self.params['x_dim']
        return self._estimate_is(x_points, sd_x, scale, sample_size, df, log)

    def get_jacobian(self, data_z):
        """Batch Jacobian of g_net at data_z, shape [batch_size, x_dim, z_dim] (roundtrip.py:312-324)."""
        data_z = np.asarray(data_z)
        assert data_z.shape[-1] == self.params['z_dim']
        return self._jacobian(data_z)[1].astype(np.float32)

    def estimate_px_with_CF(self, x_points, sd_x=0.5, log=True):
        """p(x) by the Laplace approximation (closed form), roundtrip.py:327-369."""
        x_points = np.asarray(x_points)
        assert x_points.shape[-1] == self.params['x_dim']
        return self._estimate_cf(x_points, sd_x, log)

    def estimate_py_with_IS(self, y_points, epoch=0, sd_y=0.5, scale=0.5, sample_size=40000, bs=1024, log=True,
                            save=True, df=1):
        """v1 spelling (main_density_est.py:266): y is the data point.  `save` writes
        '<save_dir>/py_est_at_epoch<epoch>.npy' like the reference (:346-347)."""
        py_est = self.estimate_px_with_IS(y_points, sd_x=sd_y, scale=scale, sample_size=sample_size, bs=bs,
                                          df=df, log=log)
        if save:
            np.save('%s/py_est_at_epoch%d.npy' % (self.save_dir, epoch), py_est)
        return py_est


class RoundtripModel(_IsModelBase):
    """v1 API (main_density_est.py:29-44).  g_net / h_net are roundtrip_amd.nets.Generator objects
    (model.Generator's constructor, model.py:54); dx_net / dy_net, the samplers and the pool are training
    parts and may be None."""

    def __init__(self, g_net, h_net, dx_net=None, dy_net=None, x_sampler=None, y_sampler=None, data='data',
                 pool=None, batch_size=64, alpha=10.0, beta=10.0, df=1, is_train=False, device=None):
        if not isinstance(g_net, DenseNet) or not isinstance(h_net, DenseNet):
            raise TypeError("g_net and h_net must be roundtrip_amd.nets.Generator / DenseNet objects")
        if g_net.input_dim != h_net.output_dim or g_net.output_dim != h_net.input_dim:
            raise ValueError("g_net maps %d->%d but h_net maps %d->%d"
                             % (g_net.input_dim, g_net.output_dim, h_net.input_dim, h_net.output_dim))
        self.data = data
        self.g_net = g_net
        self.h_net = h_net
        self.dx_net = dx_net
        self.dy_net = dy_net
        self.x_sampler = x_sampler
        self.y_sampler = y_sampler
        self.batch_size = batch_size
        self.alpha = alpha
        self.beta = beta
        self.df = df
        self.pool = pool
        self.x_dim = g_net.input_dim     # reference: self.dx_net.input_dim (:43)
        self.y_dim = g_net.output_dim    # reference: self.dy_net.input_dim (:44)
        now = datetime.datetime.now().astimezone()
        self.timestamp = now.strftime('%Y%m%d_%H%M%S')
        self.save_dir = 'data/density_est_{}_{}_x_dim={}_y_dim={}_alpha={}_beta={}'.format(
            self.timestamp, self.data, self.x_dim, self.y_dim, self.alpha, self.beta)
        if not os.path.exists(self.save_dir) and is_train:
            os.makedirs(self.save_dir)
        self._init_engine_state(device)
        self._bind()

    def _nets(self):
        return ((_lib.RT_MLP_G, self.g_net), (_lib.RT_MLP_H, self.h_net))

    def predict_y(self, x, bs=256):
        # main_density_est.py:222-234: float64 container holding float32 values
        x = np.asarray(x)
        assert x.shape[-1] == self.x_dim
        return self._forward(_lib.RT_MLP_G, np.ascontiguousarray(x, dtype=np.float32)).astype(np.float64)

    def predict_x(self, y, bs=256):
        # main_density_est.py:237-249
        y = np.asarray(y)
        assert y.shape[-1] == self.y_dim
        return self._forward(_lib.RT_MLP_H, np.ascontiguousarray(y, dtype=np.float32)).astype(np.float64)

    def estimate_py_with_IS(self, y_points, epoch, sd_y=0.5, scale=0.5, sample_size=40000, bs=1024, log=True,
                            save=True):
        """main_density_est.py:266-348; df comes from self.df (:41)."""
        y_points = np.asarray(y_points)
        assert y_points.shape[-1] == self.y_dim
        py_est = self._estimate_is(y_points, sd_y, scale, sample_size, self.df, log)
        if save:
            if not os.path.exists(self.save_dir):
                os.makedirs(self.save_dir)
            np.save('%s/py_est_at_epoch%d.npy' % (self.save_dir, epoch), py_est)
        return py_est

    def get_jacobian(self, x, bs=16):
        """(N, y_dim, x_dim) Jacobian of G (main_density_est.py:252-263); `bs` only batches the reference."""
        x = np.asarray(x)
        assert x.shape[-1] == self.x_dim
        return self._jacobian(x)[1]

    def estimate_py_with_CF(self, y_points, epoch, sd_y=0.5, log=True, save=True):
        """p(y) by the Laplace approximation (closed form), main_density_est.py:351-387."""
        y_points = np.asarray(y_points)
        assert y_points.shape[-1] == self.y_dim
        py_est = self._estimate_cf(y_points, sd_y, log)
        if save:
            if not os.path.exists(self.save_dir):
                os.makedirs(self.save_dir)
            np.save('%s/py_est_at_epoch%d.npy' % (self.save_dir, epoch), py_est)
        return py_est

    def estimate_px_with_CF(self, x_points, sd_x=0.5, log=True):
        """v2 spelling on the v1 object."""
        x_points = np.asarray(x_points)
        assert x_points.shape[-1] == self.y_dim
        return self._estimate_cf(x_points, sd_x, log)

    def model_selection(self, sample_size=20000):
        """main_density_est.py:207-219: picks (sd, scale) maximising the mean validation log-likelihood over
        sd in {.05,.1,.5} x scale in {.005,.01,.1,.5,1}.  The reference runs 15 estimates; here each scale is one
        sampler + G pass scored for the three sd values (rt_is_estimate_multi), with identical results."""
        data_y_val = self.y_sampler.X_val
        sd_list = [0.05, 0.1, 0.5]
        scale_list = [0.005, 0.01, 0.1, 0.5, 1]
        mean = {}
        for scale in scale_list:
            est = self._estimate_is_multi(data_y_val, sd_list, scale, sample_size, self.df, True)
            for sd, row in zip(sd_list, est):
                mean[(sd, scale)] = np.mean(row)
        records = []
        for sd in sd_list:
            for scale in scale_list:
                records.append([sd, scale, mean[(sd, scale)]])
        # sort according to the likelihood of validation set
        records.sort(key=lambda item: item[-1])
        best_sd, best_scale = records[-1][0], records[-1][1]
        return best_sd, best_scale

    def likelihood_summary(self, y_points, epoch, sd_y, scale, sample_size=2000, log_file=None, save=False):
        """The validation / test bookkeeping of train() (main_density_est.py:186-196): mean log-likelihood and its
        standard error, reduced on the device in numpy's summation order (rt_is_estimate_summary / rt_mean_stderr: the
        bits of np.mean(py_est) and np.std(py_est) / np.sqrt(len(py_est))); appends '<epoch>\t<mean>\t<stderr>' to
        log_file (an open file) when given."""
        y_points = np.asarray(y_points)
        assert y_points.ndim == 2 and y_points.shape[-1] == self.y_dim
        eng = self.engine
        if self.is_sampler == "philox" and not (self.shard and dist.active()) and y_points.shape[0] > 0:
            pts = np.ascontiguousarray(y_points, dtype=np.float64)
            self._sync_weights()
            centres = eng.mlp_forward(_lib.RT_MLP_H, pts.astype(np.float32))
            py_est, average_likelihood, sd_likelihood = eng.is_estimate_summary(
                pts, centres, sample_size, sd_y, scale, df=self.df, seed=self.is_seed)
            if save:
                if not os.path.exists(self.save_dir):
                    os.makedirs(self.save_dir)
                np.save('%s/py_est_at_epoch%d.npy' % (self.save_dir, epoch), py_est)
        else:
            # sharded over ranks (or the reference-draws mode): the gathered results go back to the device for the statistic
            py_est = self.estimate_py_with_IS(y_points, epoch, sd_y=sd_y, scale=scale, sample_size=sample_size,
                                              log=True, save=save)
            average_likelihood, sd_likelihood = eng.mean_stderr(py_est)
        if log_file is not None:
            log_file.write('%d\t%f\t%f\n' % (epoch, average_likelihood, sd_likelihood))
        return average_likelihood, sd_likelihood

    def estimate_px_with_IS(self, x_points, sd_x=0.5, scale=0.5, sample_size=40000, bs=1024, df=1, log=True):
        """v2 spelling on the v1 object (x is the data point there)."""
        x_points = np.asarray(x_points)
        assert x_points.shape[-1] == self.y_dim
        return self._estimate_is(x_points, sd_x, scale, sample_size, df, log)

    def load(self, pre_trained=False, timestamp='', epoch=999):
        """main_density_est.py:397-411: restores G / H from a tf.train.Saver checkpoint."""
        from . import checkpoint
        if pre_trained:
            print('Loading Pre-trained Model...')
            checkpoint_dir = 'pre_trained_models/{}'.format(self.data)
            checkpoint.load_v1(os.path.join(checkpoint_dir, 'model.ckpt-best'), self.g_net, self.h_net)
            print('Restored pre-trained model.')
        elif timestamp == '':
            print('Best Timestamp not provided.')
        else:
            checkpoint_dir = 'checkpoint/density_est_{}_{}_x_dim={}_y_dim={}_alpha={}_beta={}'.format(
                timestamp, self.data, self.x_dim, self.y_dim, self.alpha, self.beta)
            checkpoint.load_v1(os.path.join(checkpoint_dir, 'model.ckpt-%d' % epoch), self.g_net, self.h_net)
            print('Restored model weights.')
