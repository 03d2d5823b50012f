"""ctypes binding of librt_is.so (include/rt_is.h): the only compute path of this package.

There is no CPU fallback: if the library is missing or no HIP device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_IS_LIB points the binding at a diagnostic build of the same library (scripts/stamps.py); default: in-tree
LIB_PATH = os.environ.get("RT_IS_LIB") or os.path.join(_HERE, "librt_is.so")

RT_MLP_G = 0
RT_MLP_H = 1

# every symbol include/rt_is.h declares (tests check the library exports each one)
SYMBOLS = (
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_set_mlp", "rt_mlp_forward", "rt_mlp_jacobian", "rt_encode",
    "rt_is_estimate", "rt_is_estimate_multi", "rt_is_estimate_given_samples", "rt_draw_samples", "rt_is_estimate_dev",
    "rt_mlp_forward_dev", "rt_sync", "rt_get_counters", "rt_reset_counters", "rt_device_cu_count",
    "rt_is_pair_count", "rt_is_pairs", "rt_is_merge_pairs", "rt_is_estimate_dev_noalloc",
    "rt_philox4x32_10", "rt_log1p_pos", "rt_crc32c", "rt_get_xcd_shares",
    "rt_dev_alloc", "rt_dev_free", "rt_memcpy_h2d", "rt_memcpy_d2h", "rt_dev_mem_info",
    "rt_comm_unique_id", "rt_comm_init", "rt_comm_destroy", "rt_comm_info", "rt_comm_all_gather_f64",
    "rt_comm_all_gather_f64_dev", "rt_comm_all_reduce_max_f64", "rt_comm_barrier",
    "rt_is_kernel_name", "rt_cf_estimate", "rt_cf_estimate_dev", "rt_mean_stderr", "rt_mean_stderr_dev", "rt_is_estimate_summary",
)

RT_ETOOBIG = 6
RT_ECOMM = 7
RT_EBUSY = 8
RT_NOALLOC_SCRATCH_BYTES = 32 << 20
RT_COMM_ID_BYTES = 128


class RtCounters(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("flops", C.c_double), ("launches", C.c_uint64),
                ("sample_evals", C.c_uint64)]


class RtError(RuntimeError):
    def __init__(self, msg, code=None):
        RuntimeError.__init__(self, msg)
        self.code = code


_lib = None


def load():
    """Loads librt_is.so (once).  Raises ImportError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  roundtrip_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    fpp = C.POINTER(C.POINTER(C.c_float))
    lib.rt_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_ctx_destroy.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_set_mlp.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int32), fpp, fpp, C.c_float]
    lib.rt_mlp_forward.argtypes = [vp, C.c_int, vp, i64, vp]
    lib.rt_mlp_jacobian.argtypes = [vp, C.c_int, vp, i64, vp, vp]
    lib.rt_encode.argtypes = [vp, vp, i64, vp]
    lib.rt_is_estimate.argtypes = [vp, vp, vp, i64, i64, i64, dbl, dbl, dbl, C.c_uint64, C.c_int, vp]
    lib.rt_is_estimate_multi.argtypes = [vp, vp, vp, i64, i64, i64, C.c_int, vp, dbl, dbl, C.c_uint64, C.c_int, vp]
    lib.rt_is_estimate_given_samples.argtypes = [vp, vp, vp, vp, i64, i64, dbl, dbl, dbl, C.c_int, vp]
    lib.rt_draw_samples.argtypes = [vp, vp, i64, i64, i64, dbl, dbl, C.c_uint64, vp]
    lib.rt_is_estimate_dev.argtypes = [vp, vp, vp, i64, i64, i64, dbl, dbl, dbl, C.c_uint64, C.c_int, vp]
    lib.rt_mlp_forward_dev.argtypes = [vp, C.c_int, vp, i64, vp]
    lib.rt_sync.argtypes = [vp]
    lib.rt_get_counters.argtypes = [vp, C.POINTER(RtCounters)]
    lib.rt_reset_counters.argtypes = [vp]
    lib.rt_device_cu_count.argtypes = [vp, C.POINTER(C.c_int)]
    lib.rt_is_pair_count.argtypes = [vp, i64, C.POINTER(i64)]
    lib.rt_is_pairs.argtypes = [vp, vp, vp, i64, i64, i64, C.c_int, vp, dbl, dbl, C.c_uint64, i64, i64, vp]
    lib.rt_is_merge_pairs.argtypes = [vp, vp, i64, i64, C.c_int, i64, C.c_int, vp]
    lib.rt_is_estimate_dev_noalloc.argtypes = lib.rt_is_estimate_dev.argtypes
    lib.rt_philox4x32_10.argtypes = [vp, vp, i64, vp]
    lib.rt_log1p_pos.argtypes = [vp, vp, i64, vp]
    lib.rt_get_xcd_shares.argtypes = [vp, C.POINTER(C.c_double)]
    lib.rt_crc32c.restype = C.c_uint32
    lib.rt_crc32c.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    lib.rt_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.rt_dev_free.argtypes = [vp, vp]
    lib.rt_dev_mem_info.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.rt_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.rt_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.rt_comm_unique_id.argtypes = [vp]
    lib.rt_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.rt_comm_destroy.argtypes = [vp]
    lib.rt_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rt_comm_all_gather_f64.argtypes = [vp, vp, i64, vp]
    lib.rt_comm_all_gather_f64_dev.argtypes = [vp, vp, i64, vp]
    lib.rt_comm_all_reduce_max_f64.argtypes = [vp, vp, i64]
    lib.rt_comm_barrier.argtypes = [vp]
    lib.rt_is_kernel_name.argtypes = [vp, dbl, C.c_char_p, C.c_size_t]
    lib.rt_cf_estimate.argtypes = [vp, vp, i64, dbl, C.c_int, vp]
    lib.rt_cf_estimate_dev.argtypes = [vp, vp, i64, dbl, C.c_int, vp]
    lib.rt_mean_stderr.argtypes = [vp, vp, i64, C.POINTER(dbl), C.POINTER(dbl)]
    lib.rt_mean_stderr_dev.argtypes = [vp, vp, i64, vp]
    lib.rt_is_estimate_summary.argtypes = [vp, vp, vp, i64, i64, i64, dbl, dbl, dbl, C.c_uint64, vp, C.POINTER(dbl),
                                           C.POINTER(dbl)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("rt_last_error", "rt_crc32c"):
            fn.restype = C.c_int
    _lib = lib
    return lib


class Engine(object):
    """One rt_ctx: a HIP device, a stream, and the uploaded G / H networks."""

    def __init__(self, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.rt_ctx_create(int(device), C.byref(h))
        if rc:
            raise RtError("rt_ctx_create(device=%d) failed (%d): %s"
                          % (device, rc, self._lib.rt_last_error(None).decode()))
        self._h = h
        self.device = int(device)
        self.dims = {}

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rt_ctx_destroy(h)

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise RtError("librt_is error %d: %s" % (rc, self._lib.rt_last_error(self._h).decode()), rc)

    def set_mlp(self, which, weights, biases, alpha=0.2):
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, dtype=np.float32).reshape(-1) for b in biases]
        if len(ws) != len(bs) or not ws:
            raise ValueError("weights and biases must be non-empty lists of equal length")
        dims = [ws[0].shape[0]]
        for w, b in zip(ws, bs):
            if w.ndim != 2 or w.shape[0] != dims[-1] or b.shape[0] != w.shape[1]:
                raise ValueError("layer shapes do not chain: %r / %r after width %d" % (w.shape, b.shape, dims[-1]))
            dims.append(w.shape[1])
        n = len(ws)
        fp = C.POINTER(C.c_float)
        W = (fp * n)(*[w.ctypes.data_as(fp) for w in ws])
        B = (fp * n)(*[b.ctypes.data_as(fp) for b in bs])
        D = (C.c_int32 * (n + 1))(*dims)
        self._chk(self._lib.rt_set_mlp(self._h, int(which), n, D, W, B, float(alpha)))
        self.dims[int(which)] = dims

    def mlp_forward(self, which, x):
        dims = self.dims.get(int(which))
        if dims is None:
            raise RtError("MLP %d has not been set" % which)
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != dims[0]:
            raise ValueError("expected shape (N, %d), got %r" % (dims[0], x.shape))
        out = np.empty((x.shape[0], dims[-1]), dtype=np.float32)
        self._chk(self._lib.rt_mlp_forward(self._h, int(which), x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def mlp_jacobian(self, which, x):
        """(MLP(x) (N, out), J (N, out, in)) float32: get_jacobian (main_density_est.py:252-263) on the GPU."""
        dims = self.dims.get(int(which))
        if dims is None:
            raise RtError("MLP %d has not been set" % which)
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != dims[0]:
            raise ValueError("expected shape (N, %d), got %r" % (dims[0], x.shape))
        out = np.empty((x.shape[0], dims[-1]), dtype=np.float32)
        J = np.empty((x.shape[0], dims[-1], dims[0]), dtype=np.float32)
        self._chk(self._lib.rt_mlp_jacobian(self._h, int(which), x.ctypes.data, x.shape[0], out.ctypes.data,
                                            J.ctypes.data))
        return out, J

    def _g_dims(self):
        dims = self.dims.get(RT_MLP_G)
        if dims is None:
            raise RtError("G has not been set")
        return dims[0], dims[-1]

    def _points(self, y, xhat):
        """(y float64 (N, dy), xhat float32 (N, dx), N, dx), contiguous and shape-checked against the G that is set."""
        dx, dy = self._g_dims()
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.ndim != 2 or y.shape[1] != dy:
            raise ValueError("expected y of shape (N, %d), got %r" % (dy, y.shape))
        N = y.shape[0]
        xhat = np.ascontiguousarray(xhat, dtype=np.float32)
        if xhat.shape != (N, dx):
            raise ValueError("expected xhat of shape (%d, %d), got %r" % (N, dx, xhat.shape))
        return y, xhat, N, dx

    @staticmethod
    def _sds(sd_ys):
        """The sd_y values of a several-sd call as a flat float64 array."""
        return np.ascontiguousarray(sd_ys, dtype=np.float64).reshape(-1)

    def is_estimate(self, y, xhat, sample_size, sd_y, scale, df=1.0, seed=0, log=True, y_index_base=0):
        y, xhat, N, dx = self._points(y, xhat)
        out = np.empty(N, dtype=np.float64)
        self._chk(self._lib.rt_is_estimate(self._h, y.ctypes.data, xhat.ctypes.data, N, int(y_index_base),
                                           int(sample_size), float(sd_y), float(scale), float(df),
                                           int(seed), int(bool(log)), out.ctypes.data))
        return out

    def is_estimate_summary(self, y, xhat, sample_size, sd_y, scale, df=1.0, seed=0, y_index_base=0):
        """(log p(y) (N,), mean, standard error): rt_is_estimate with log=True and the statistic train() keeps of it
        (main_density_est.py:186-188), reduced on the device in numpy's summation order."""
        y, xhat, N, dx = self._points(y, xhat)
        out = np.empty(N, dtype=np.float64)
        mean, se = C.c_double(), C.c_double()
        self._chk(self._lib.rt_is_estimate_summary(self._h, y.ctypes.data, xhat.ctypes.data, N, int(y_index_base),
                                                   int(sample_size), float(sd_y), float(scale), float(df), int(seed),
                                                   out.ctypes.data, C.byref(mean), C.byref(se)))
        return out, np.float64(mean.value), np.float64(se.value)

    def mean_stderr(self, values):
        """(np.mean(values), np.std(values) / np.sqrt(len(values))) computed on the device (rt_mean_stderr)."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        mean, se = C.c_double(), C.c_double()
        self._chk(self._lib.rt_mean_stderr(self._h, v.ctypes.data, v.size, C.byref(mean), C.byref(se)))
        return np.float64(mean.value), np.float64(se.value)

    def mean_stderr_dev(self, values_ptr, N, out2_ptr):
        self._chk(self._lib.rt_mean_stderr_dev(self._h, values_ptr, int(N), out2_ptr))

    def cf_estimate(self, y, sd_y, log=True):
        """Laplace (closed-form) estimate for every row of y (N, dy): rt_cf_estimate, everything on the device."""
        _, dy = self._g_dims()
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.ndim != 2 or y.shape[1] != dy:
            raise ValueError("expected y of shape (N, %d), got %r" % (dy, y.shape))
        out = np.empty(y.shape[0], dtype=np.float64)
        self._chk(self._lib.rt_cf_estimate(self._h, y.ctypes.data, y.shape[0], float(sd_y), int(bool(log)),
                                           out.ctypes.data))
        return out

    def cf_estimate_dev(self, y_ptr, N, sd_y, out_ptr, log=True):
        self._chk(self._lib.rt_cf_estimate_dev(self._h, y_ptr, int(N), float(sd_y), int(bool(log)), out_ptr))

    def is_estimate_multi(self, y, xhat, sample_size, sd_ys, scale, df=1.0, seed=0, log=True, y_index_base=0):
        """One sampler + G pass scored for every sd in sd_ys; returns (len(sd_ys), N)."""
        y, xhat, N, dx = self._points(y, xhat)
        sds = self._sds(sd_ys)
        out = np.empty((sds.size, N), dtype=np.float64)
        self._chk(self._lib.rt_is_estimate_multi(self._h, y.ctypes.data, xhat.ctypes.data, N, int(y_index_base),
                                                 int(sample_size), int(sds.size), sds.ctypes.data, float(scale),
                                                 float(df), int(seed), int(bool(log)), out.ctypes.data))
        return out

    # ---- one query point's samples split over callers (rt_is.h: pairs) --------------------------------------
    def is_pair_count(self, sample_size):
        """E: (max, sum) pairs per query point and sd value for this sample_size under the G that is set."""
        e = C.c_int64(0)
        self._chk(self._lib.rt_is_pair_count(self._h, C.c_int64(int(sample_size)), C.byref(e)))
        return int(e.value)

    def is_pairs(self, y, xhat, sample_size, sd_ys, scale, e_lo, e_hi, df=1.0, seed=0, y_index_base=0):
        """Pairs [e_lo, e_hi) of every query point: (N, e_hi - e_lo, len(sd_ys), 2) float64."""
        y, xhat, N, dx = self._points(y, xhat)
        sds = self._sds(sd_ys)
        out = np.empty((N, max(0, int(e_hi) - int(e_lo)), sds.size, 2), dtype=np.float64)
        self._chk(self._lib.rt_is_pairs(self._h, y.ctypes.data, xhat.ctypes.data, C.c_int64(N),
                                        C.c_int64(int(y_index_base)), C.c_int64(int(sample_size)), int(sds.size),
                                        sds.ctypes.data, C.c_double(float(scale)), C.c_double(float(df)),
                                        C.c_uint64(int(seed)), C.c_int64(int(e_lo)), C.c_int64(int(e_hi)),
                                        out.ctypes.data))
        return out

    def is_merge_pairs(self, pairs, sample_size, log=True):
        """(n_sd, N) estimates from all E pairs of every point: pairs is (N, E, n_sd, 2)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.float64)
        if pairs.ndim != 4 or pairs.shape[3] != 2 or pairs.shape[1] < 1:
            raise ValueError("expected pairs of shape (N, E, n_sd, 2), got %r" % (pairs.shape,))
        N, E, n_sd, _ = pairs.shape
        out = np.empty((n_sd, N), dtype=np.float64)
        self._chk(self._lib.rt_is_merge_pairs(self._h, pairs.ctypes.data, C.c_int64(N), C.c_int64(E), int(n_sd),
                                              C.c_int64(int(sample_size)), int(bool(log)), out.ctypes.data))
        return out

    def is_estimate_given_samples(self, y, xhat, x_samples, sd_y, scale, df=1.0, log=True):
        y, xhat, N, dx = self._points(y, xhat)
        xs = np.ascontiguousarray(x_samples, dtype=np.float64)
        if xs.ndim != 3 or xs.shape[0] != N or xs.shape[2] != dx or xs.shape[1] < 1:
            raise ValueError("expected x_samples of shape (%d, S, %d), got %r" % (N, dx, xs.shape))
        out = np.empty(N, dtype=np.float64)
        self._chk(self._lib.rt_is_estimate_given_samples(self._h, y.ctypes.data, xhat.ctypes.data,
                                                         xs.ctypes.data, N, xs.shape[1], float(sd_y),
                                                         float(scale), float(df), int(bool(log)),
                                                         out.ctypes.data))
        return out

    def draw_samples(self, xhat, sample_size, scale, df=1.0, seed=0, y_index_base=0):
        dx, _ = self._g_dims()
        xhat = np.ascontiguousarray(xhat, dtype=np.float32)
        if xhat.ndim != 2 or xhat.shape[1] != dx:
            raise ValueError("expected xhat of shape (N, %d), got %r" % (dx, xhat.shape))
        xs = np.empty((xhat.shape[0], int(sample_size), dx), dtype=np.float32)
        self._chk(self._lib.rt_draw_samples(self._h, xhat.ctypes.data, xhat.shape[0], int(y_index_base),
                                            int(sample_size), float(scale), float(df), int(seed),
                                            xs.ctypes.data))
        return xs

    def philox4x32_10(self, ctr_key):
        """(n, 4) uint32 outputs of the sampler's generator for (n, 6) uint32 rows (counter[4], key[2])."""
        ck = np.ascontiguousarray(ctr_key, dtype=np.uint32).reshape(-1, 6)
        out = np.empty((ck.shape[0], 4), dtype=np.uint32)
        self._chk(self._lib.rt_philox4x32_10(self._h, ck.ctypes.data, ck.shape[0], out.ctypes.data))
        return out

    def log1p_pos(self, a):
        """log(1 + a), a >= 0, as the scorer evaluates it on the device (diagnostic, for accuracy tests)."""
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        out = np.empty_like(a)
        self._chk(self._lib.rt_log1p_pos(self._h, a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    # ---- device-resident entry points (pointers are raw device addresses, e.g. torch .data_ptr()) ----
    def is_estimate_dev(self, y_ptr, xhat_ptr, N, sample_size, sd_y, scale, out_ptr, df=1.0, seed=0,
                        log=True, y_index_base=0):
        self._chk(self._lib.rt_is_estimate_dev(self._h, y_ptr, xhat_ptr, int(N), int(y_index_base),
                                               int(sample_size), float(sd_y), float(scale), float(df),
                                               int(seed), int(bool(log)), out_ptr))

    def is_estimate_dev_noalloc(self, y_ptr, xhat_ptr, N, sample_size, sd_y, scale, out_ptr, df=1.0, seed=0,
                                log=True, y_index_base=0):
        """rt_is_estimate_dev for callers that cannot allocate (rt_is.h): pairs go to the per-GPU scratch array."""
        self._chk(self._lib.rt_is_estimate_dev_noalloc(self._h, y_ptr, xhat_ptr, int(N), int(y_index_base),
                                                       int(sample_size), float(sd_y), float(scale), float(df),
                                                       int(seed), int(bool(log)), out_ptr))

    # ---- device memory (rt_dev_alloc & co.): what callers of the *_dev entry points keep their arrays in ----
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self._lib.rt_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        if ptr and getattr(self, "_h", None):
            self._chk(self._lib.rt_dev_free(self._h, ptr))

    def dev_mem_info(self):
        """(free, total) bytes of device memory on the engine's GPU."""
        f, t = C.c_size_t(), C.c_size_t()
        self._chk(self._lib.rt_dev_mem_info(self._h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def to_device(self, arr):
        """A fresh device copy of a numpy array; returns the raw device address (free it with dev_free)."""
        arr = np.ascontiguousarray(arr)
        ptr = self.dev_alloc(arr.nbytes)
        self._chk(self._lib.rt_memcpy_h2d(self._h, ptr, arr.ctypes.data, arr.nbytes))
        return ptr

    def from_device(self, ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._chk(self._lib.rt_memcpy_d2h(self._h, out.ctypes.data, ptr, out.nbytes))
        return out

    # ---- the final gather over GPUs: RCCL behind the C ABI (rt_comm_*) ----
    def comm_unique_id(self):
        buf = C.create_string_buffer(RT_COMM_ID_BYTES)
        rc = self._lib.rt_comm_unique_id(buf)
        if rc:
            raise RtError("rt_comm_unique_id failed (%d): %s" % (rc, self._lib.rt_last_error(None).decode()), rc)
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        if len(unique_id) != RT_COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % RT_COMM_ID_BYTES)
        self._chk(self._lib.rt_comm_init(self._h, int(nranks), int(rank), C.c_char_p(bytes(unique_id))))

    def comm_destroy(self):
        self._chk(self._lib.rt_comm_destroy(self._h))

    def comm_info(self):
        r, n = C.c_int(), C.c_int()
        self._chk(self._lib.rt_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_all_gather(self, local):
        """(nranks, len(local)) float64: every rank's `local` (all of one length), through RCCL."""
        local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
        out = np.empty((self.comm_info()[1], local.size), dtype=np.float64)
        self._chk(self._lib.rt_comm_all_gather_f64(self._h, local.ctypes.data, local.size, out.ctypes.data))
        return out

    def comm_all_gather_dev(self, send_ptr, count, recv_ptr):
        self._chk(self._lib.rt_comm_all_gather_f64_dev(self._h, send_ptr, int(count), recv_ptr))

    def comm_all_reduce_max(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()
        self._chk(self._lib.rt_comm_all_reduce_max_f64(self._h, v.ctypes.data, v.size))
        return v

    def comm_barrier(self):
        self._chk(self._lib.rt_comm_barrier(self._h))

    def mlp_forward_dev(self, which, in_ptr, N, out_ptr):
        self._chk(self._lib.rt_mlp_forward_dev(self._h, int(which), in_ptr, int(N), out_ptr))

    def sync(self):
        self._chk(self._lib.rt_sync(self._h))

    def counters(self):
        c = RtCounters()
        self._chk(self._lib.rt_get_counters(self._h, C.byref(c)))
        return {"kernel_ms": c.kernel_ms, "flops": c.flops, "launches": int(c.launches),
                "sample_evals": int(c.sample_evals)}

    def reset_counters(self):
        self._chk(self._lib.rt_reset_counters(self._h))

    def xcd_shares(self):
        """Shares of the fused kernel's passes the eight XCD classes get in the next chip-filling launch (rt_is.h)."""
        a = (C.c_double * 8)()
        self._chk(self._lib.rt_get_xcd_shares(self._h, a))
        return np.array(list(a))

    def is_kernel_name(self, df=1.0):
        """The fused kernel's instantiation for the G that is set, as a kernel trace names it."""
        buf = C.create_string_buffer(128)
        self._chk(self._lib.rt_is_kernel_name(self._h, float(df), buf, 128))
        return buf.value.decode()

    def cu_count(self):
        n = C.c_int()
        self._chk(self._lib.rt_device_cu_count(self._h, C.byref(n)))
        return n.value
