"""numpy/scipy restatement of Roundtrip's importance-sampling density estimator.

TEST INFRASTRUCTURE ONLY: imported by tests/, oracle/make_golden.py, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under roundtrip_amd/ imports it.

PARITY UNPINNED BY THE REFERENCE: the reference holds no tests or golden vectors for this path
 and its TensorFlow model cannot be imported here.  What pins this file is that it
calls the reference's own third-party arithmetic — numpy's legacy global RandomState (seed, chisquare,
multivariate_normal) and scipy.stats.t.logpdf — statement by statement, single-threaded so the draw
order is defined; only the TensorFlow dense layers are restated (float32 `x @ W + b`, leaky-ReLU).

Follows /root/reference/main_density_est.py:266-348 (v2 twin: src/pyroundtrip/roundtrip.py:229-310) and, for the
Laplace estimator, :252-263 and :351-387 (v2: roundtrip.py:312-369).
"""
import numpy as np
from scipy.stats import t


def leaky_relu(x, alpha=0.2):
    # model.py:5-6  tf.maximum(tf.minimum(0.0, alpha * x), x)
    return np.maximum(np.minimum(np.float32(0.0), np.float32(alpha) * x), x)


class DenseNet(object):
    """model.Generator.__call__ (model.py:61-88) / BaseFullyConnectedNet.call
    (src/pyroundtrip/model.py:29-50): nb_layers x (fully_connected + leaky_relu), linear output layer.
    Weights are [in, out]; computation is float32."""

    def __init__(self, weights, biases, alpha=0.2):
        self.weights = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self.biases = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        self.alpha = alpha
        self.input_dim = self.weights[0].shape[0]
        self.output_dim = self.weights[-1].shape[1]

    def __call__(self, x):
        fc = np.asarray(x, dtype=np.float32)  # float32 placeholder feed, main_density_est.py:232
        for w, b in zip(self.weights[:-1], self.biases[:-1]):
            fc = leaky_relu(fc @ w + b, self.alpha)
        return fc @ self.weights[-1] + self.biases[-1]


def glorot_mlp(dims, rng):
    """Glorot-uniform weights (tcl.fully_connected's default initializer), zero biases.
    `rng` is a np.random.RandomState; BASELINE.md §4 fixes RandomState(1234)."""
    ws, bs = [], []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        ws.append(rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32))
        bs.append(np.zeros(fan_out, dtype=np.float32))
    return ws, bs


def predict(net, x, out_dim, bs=256):
    # predict_y / predict_x, main_density_est.py:222-249: float64 container holding float32 results
    N = x.shape[0]
    pred = np.zeros(shape=(N, out_dim))
    for b in range(int(np.ceil(N * 1.0 / bs))):
        ind = np.arange(b * bs, min((b + 1) * bs, N))
        pred[ind, :] = net(x[ind, :])
    return pred


def sample_from_qx(x_point, x_dim, scale, df, sample_size):
    # main_density_est.py:310-318 — uses the GLOBAL legacy numpy RNG, like the reference
    S = np.diag(scale ** 2 * np.ones(x_dim))
    z1 = np.random.chisquare(df, sample_size) / df
    z2 = np.random.multivariate_normal(np.zeros(x_dim), S, (sample_size,))
    return x_point + z2 / np.sqrt(z1)[:, None]


def py_given_x(g_net, x_points, y_point, y_dim, sd_y, bs, log):
    # main_density_est.py:278-290
    y_points_ = predict(g_net, x_points, y_dim, bs=bs)
    if log:
        return -y_dim * np.log((np.sqrt(2 * np.pi) * sd_y)) - (np.sum((y_point - y_points_) ** 2, axis=1)) / (2. * sd_y ** 2)
    return 1. / ((np.sqrt(2 * np.pi) * sd_y) ** y_dim) * np.exp(-(np.sum((y_point - y_points_) ** 2, axis=1)) / (2. * sd_y ** 2))


def w_likelihood_ratio(x_point, x_points, x_dim, scale, df, log):
    # main_density_est.py:292-307
    if log:
        log_qx = np.sum(t.logpdf(x_point - x_points, df, loc=0, scale=scale), axis=1)
        log_px = -x_dim * np.log(np.sqrt(2 * np.pi)) - (np.sum((x_points) ** 2, axis=1)) / 2.
        return log_px - log_qx
    qx = np.prod(t.pdf(x_point - x_points, df, loc=0, scale=scale), axis=1)
    px = 1. / (np.sqrt(2 * np.pi) ** x_dim) * np.exp(-(np.sum((x_points) ** 2, axis=1)) / 2.)
    return px / qx


def reduce_point(a, b, sample_size, log):
    # main_density_est.py:339-345 for one query point
    if log:
        item = a + b
        idx = np.where(item == max(item))[0][0]
        return np.log(np.sum(np.exp(item - item[idx]))) - np.log(sample_size) + item[idx]
    return np.mean(a * b)


def estimate_py_with_IS(g_net, h_net, y_points, sd_y=0.5, scale=0.5, sample_size=40000, bs=1024, df=1,
                        log=True, return_samples=False, x_samples=None):
    """The reference estimator, one query point after another (single-threaded, so the global-RNG
    draw order is defined: point 0's chisquare, point 0's normals, point 1's chisquare, ...).

    x_samples: optional (N, S, x_dim) float64 array used INSTEAD of drawing (the RNG is then untouched).
    Returns py_est (N,) float64 [, x_samples (N, S, x_dim), x_points_ (N, x_dim)].
    """
    np.random.seed(0)  # :267
    x_dim, y_dim = g_net.input_dim, g_net.output_dim
    y_points = np.asarray(y_points, dtype=np.float64)
    x_points_ = predict(h_net, y_points, x_dim, bs=bs)  # :320
    N = len(y_points)
    py_est = np.zeros(N)
    kept = []
    for n in range(N):
        if x_samples is None:
            xs = sample_from_qx(x_points_[n], x_dim, scale, df, sample_size)  # :331
        else:
            xs = np.asarray(x_samples[n], dtype=np.float64)
        a = py_given_x(g_net, xs, y_points[n], y_dim, sd_y, bs, log)  # :332
        b = w_likelihood_ratio(x_points_[n], xs, x_dim, scale, df, log)  # :333
        py_est[n] = reduce_point(a, b, sample_size, log)
        if return_samples:
            kept.append(xs)
    if return_samples:
        return py_est, np.stack(kept), x_points_
    return py_est


def get_jacobian(net, x):
    """get_jacobian (main_density_est.py:252-263; self.J = batch_jacobian(self.y_, self.x), :65): (N, y_dim, x_dim)
    float64 container holding float32 values.  TensorFlow's autodiff is restated as float32 forward mode: the
    tangent of a dense layer is T @ W, and leaky_relu'(t) is 1 for t > 0, alpha otherwise."""
    fc = np.asarray(x, dtype=np.float32)
    N, x_dim = fc.shape
    tan = np.broadcast_to(np.eye(x_dim, dtype=np.float32), (N, x_dim, x_dim)).copy()   # [n, direction, feature]
    for w, b in zip(net.weights[:-1], net.biases[:-1]):
        pre = fc @ w + b
        tan = (tan @ w) * np.where(pre > 0, np.float32(1.0), np.float32(net.alpha))[:, None, :]
        fc = leaky_relu(pre, net.alpha)
    tan = tan @ net.weights[-1]
    jcob_pred = np.zeros(shape=(N, net.output_dim, x_dim))
    jcob_pred[:] = tan.transpose((0, 2, 1))
    return jcob_pred


def estimate_py_with_CF(g_net, h_net, y_points, sd_y=0.5, log=True):
    """Laplace / closed-form estimator, main_density_est.py:351-387 statement by statement (v2 twin:
    src/pyroundtrip/roundtrip.py:327-369).  Note the reference's log=False branch carries an extra sd_y**y_dim
    factor that its log=True branch does not (:382 vs :384); both are reproduced as written."""
    x_dim, y_dim = g_net.input_dim, g_net.output_dim
    y_points = np.asarray(y_points, dtype=np.float64)
    x_points_ = predict(h_net, y_points, x_dim)              # :363
    y_points__ = predict(g_net, x_points_, y_dim)            # :364
    rt_error = np.sum((y_points - y_points__) ** 2, axis=1)  # :365
    jacob_mat = get_jacobian(g_net, x_points_)               # :367
    jacob_mat_transpose = jacob_mat.transpose((0, 2, 1))     # :369
    A = list(map(lambda x, y: np.dot(x, y), jacob_mat_transpose, jacob_mat))                 # :371
    b = list(map(lambda x, y: np.dot(x, y), jacob_mat_transpose, y_points - y_points__))     # :373
    Sigma = list(map(lambda x: np.linalg.inv(np.eye(x_dim) + x / sd_y ** 2), A))             # :375
    Sigma_inv = list(map(lambda x: np.eye(x_dim) + x / sd_y ** 2, A))                        # :376
    mu = list(map(lambda x, y, z: x.dot(y / sd_y ** 2 - z), Sigma, b, x_points_))            # :378
    c_y = list(map(lambda x, y, z, w: np.sum(x ** 2) + y / sd_y ** 2 - z.T.dot(w).dot(z),
                   x_points_, rt_error, mu, Sigma_inv))                                      # :380
    if log:
        py_est = list(map(lambda x, y: -y_dim * np.log(np.sqrt(2 * np.pi) * sd_y) + 0.5 * np.log(np.linalg.det(x))
                          - 0.5 * y, Sigma, c_y))                                            # :382
    else:
        py_est = list(map(lambda x, y: 1. / (np.sqrt(2 * np.pi) * sd_y) ** y_dim * sd_y ** y_dim
                          * np.sqrt(np.linalg.det(x)) * np.exp(-0.5 * y), Sigma, c_y))       # :384
    return np.array(py_est)


def create_2d_grid_data(x1_min, x1_max, x2_min, x2_max, n=100):
    # util.py:568-573 / src/pyroundtrip/util.py:624-629 (restated: three numpy calls)
    grid_x1 = np.linspace(x1_min, x1_max, n)
    grid_x2 = np.linspace(x2_min, x2_max, n)
    v1, v2 = np.meshgrid(grid_x1, grid_x2)
    data_grid = np.vstack((v1.ravel(), v2.ravel())).T
    return v1, v2, data_grid
