"""ctypes binding of oracle/librt_is_oracle.so (built from is_oracle.c by oracle/Makefile).

TEST INFRASTRUCTURE ONLY — see the header of is_oracle.c.  Nothing under roundtrip_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librt_is_oracle.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    fpp = C.POINTER(C.POINTER(C.c_float))
    lib.rto_mlp_create.restype = C.c_void_p
    lib.rto_mlp_create.argtypes = [C.c_int, C.POINTER(C.c_int32), fpp, fpp, C.c_float]
    lib.rto_mlp_destroy.argtypes = [C.c_void_p]
    lib.rto_mlp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int]
    lib.rto_mlp_jacobian.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.rto_mlp_jacobian.restype = None
    lib.rto_t_logpdf.restype = C.c_double
    lib.rto_t_logpdf.argtypes = [C.c_double, C.c_double, C.c_double]
    lib.rto_is_estimate_given_samples.restype = C.c_int
    lib.rto_is_estimate_given_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                  C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int,
                                                  C.c_int, C.c_int, C.c_void_p]
    lib.rto_philox4x32_10.restype = None
    lib.rto_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rto_draw_samples.restype = C.c_int
    lib.rto_draw_samples.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double,
                                     C.c_double, C.c_uint64, C.c_void_p]
    lib.rto_max_threads.restype = C.c_int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def max_threads():
    return int(lib().rto_max_threads())


def t_logpdf(x, df, scale):
    return float(lib().rto_t_logpdf(float(x), float(df), float(scale)))


ORDER_NATURAL = 0   # k ascending
ORDER_KERNEL = 1    # summation order of the HIP kernel (bit-exact comparisons)


class OracleMlp(object):
    def __init__(self, weights, biases, alpha=0.2):
        self.weights = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self.biases = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        n = len(self.weights)
        dims = [self.weights[0].shape[0]] + [w.shape[1] for w in self.weights]
        self.dims = dims
        self.input_dim, self.output_dim = dims[0], dims[-1]
        fp = C.POINTER(C.c_float)
        W = (fp * n)(*[w.ctypes.data_as(fp) for w in self.weights])
        B = (fp * n)(*[b.ctypes.data_as(fp) for b in self.biases])
        D = (C.c_int32 * (n + 1))(*dims)
        self._h = lib().rto_mlp_create(n, D, W, B, float(alpha))
        if not self._h:
            raise RuntimeError("rto_mlp_create failed")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.rto_mlp_destroy(h)

    def forward(self, x, order=ORDER_NATURAL, nthreads=1):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.input_dim)
        out = np.empty((x.shape[0], self.output_dim), dtype=np.float32)
        lib().rto_mlp_forward(self._h, x.ctypes.data, x.shape[0], out.ctypes.data, int(order), int(nthreads))
        return out

    def jacobian(self, x, order=ORDER_NATURAL, nthreads=1):
        """(out (N, out_dim), J (N, out_dim, in_dim)) float32 — get_jacobian, main_density_est.py:252-263."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.input_dim)
        out = np.empty((x.shape[0], self.output_dim), dtype=np.float32)
        J = np.empty((x.shape[0], self.output_dim, self.input_dim), dtype=np.float32)
        lib().rto_mlp_jacobian(self._h, x.ctypes.data, x.shape[0], out.ctypes.data, J.ctypes.data, int(order),
                               int(nthreads))
        return out, J


def is_estimate_given_samples(g, y, xhat, xs, sd_y, scale, df=1.0, log=True, order=ORDER_NATURAL, nthreads=1):
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1, g.output_dim)
    N = y.shape[0]
    xhat = np.ascontiguousarray(xhat, dtype=np.float32).reshape(N, g.input_dim)
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    S = xs.size // (N * g.input_dim)
    assert xs.size == N * S * g.input_dim
    out = np.empty(N, dtype=np.float64)
    rc = lib().rto_is_estimate_given_samples(g._h, y.ctypes.data, xhat.ctypes.data, xs.ctypes.data, N, S,
                                             float(sd_y), float(scale), float(df), int(bool(log)), int(order),
                                             int(nthreads), out.ctypes.data)
    if rc:
        raise RuntimeError("rto_is_estimate_given_samples failed: %d" % rc)
    return out


def draw_samples(xhat, y_index_base, S, scale, df, seed):
    xhat = np.ascontiguousarray(xhat, dtype=np.float32)
    N, dx = xhat.shape
    xs = np.empty((N, S, dx), dtype=np.float32)
    rc = lib().rto_draw_samples(xhat.ctypes.data, N, int(y_index_base), int(S), dx, float(scale), float(df),
                                int(seed), xs.ctypes.data)
    if rc:
        raise RuntimeError("rto_draw_samples: unsupported df")
    return xs


def philox4x32_10(ctr, key):
    """The four output words of Philox4x32-10 for one (counter[4], key[2])."""
    c = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(4)
    k = np.ascontiguousarray(key, dtype=np.uint32).reshape(2)
    out = np.empty(4, dtype=np.uint32)
    lib().rto_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out
