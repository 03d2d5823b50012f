/*
 * rt_is.h — C ABI of librt_is.so, the MI355X (gfx950) implementation of Roundtrip's
 * importance-sampling (IS) density evaluation path.
 *
 * The reference has no FFI for this path: it is a Python method on the model object
 * (`RoundtripModel.estimate_py_with_IS`, /root/reference/main_density_est.py:266-348; v2 twin
 * `Roundtrip.estimate_px_with_IS`, /root/reference/src/pyroundtrip/roundtrip.py:229-310).  The entry
 * points below are what a ctypes binding inside that method would call; each one cites the reference
 * lines it replaces.  The binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an int status: 0 = RT_OK, otherwise an RT_E* code; the text of the last
 *     failure on a context is available from rt_last_error().  No C++ exception crosses this boundary.
 *   - all matrices are dense, row-major, little-endian.  Dense-layer weights are stored [in, out]
 *     (the TensorFlow convention: out = in @ W + b).
 *   - names follow the v1 reference: y = data point (dim dy), x = latent (dim dx), G : x -> y is MLP 0,
 *     H : y -> x is MLP 1.  (v2 spells the same things x / z / g_net / e_net.)
 *   - a context is bound to one HIP device and owns one stream.  It is not re-entrant; different
 *     contexts may be used from different threads.
 *   - pointers are HOST pointers unless the function name ends in `_dev`.
 */
#ifndef RT_IS_H
#define RT_IS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK            0
#define RT_EINVAL        1   /* bad argument (shape, NULL pointer, unsupported value) */
#define RT_EHIP          2   /* a HIP runtime call failed */
#define RT_ENOTSET       3   /* the MLP needed by the call has not been set */
#define RT_EUNSUPPORTED  4   /* shape outside what the kernels handle */
#define RT_ENOMEM        5
#define RT_ETOOBIG       6   /* the call needs more than a fixed-size buffer holds (rt_is_estimate_dev_noalloc) */
#define RT_ECOMM         7   /* RCCL could not be loaded, or one of its calls failed */
#define RT_EBUSY         8   /* a per-GPU resource is in use by another context (rt_is_estimate_dev_noalloc's scratch array) */

#define RT_MLP_G 0           /* generator  G: x (dx) -> y (dy)  : g_net, main_density_est.py:447 */
#define RT_MLP_H 1           /* encoder    H: y (dy) -> x (dx)  : h_net / e_net, main_density_est.py:448 */

#define RT_MAX_LAYERS 24
#define RT_MAX_SD 8          /* sd_y values one rt_is_estimate_multi call can score */
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)   /* the scratch array of rt_is_estimate_dev_noalloc, per GPU */
#define RT_COMM_ID_BYTES 128 /* size of the rendezvous token of rt_comm_unique_id / rt_comm_init */

typedef struct rt_ctx rt_ctx;

typedef struct rt_counters {
    double   kernel_ms;      /* sum of the fused IS kernel's durations (HIP events on the context's stream) */
    double   flops;          /* algorithmic flops of those launches: N*S*2*sum(in_l*out_l) over G's layers, unpadded */
    uint64_t launches;       /* number of fused IS kernel launches summed */
    uint64_t sample_evals;   /* N*S summed */
} rt_counters;

/* Creates a context on HIP device `device` (replaces nothing in the reference: TF session creation,
 * main_density_est.py:114-121, is the nearest analogue). */
int rt_ctx_create(int device, rt_ctx** out);
int rt_ctx_destroy(rt_ctx* ctx);

/* Text of the last error on this context ("" if none).  ctx == NULL returns the last creation error. */
const char* rt_last_error(const rt_ctx* ctx);

/* Uploads one MLP (replaces the weights `tf.train.Saver.restore` puts into the graph,
 * main_density_est.py:397-411, and the layer structure of model.Generator.__call__, model.py:61-88 /
 * BaseFullyConnectedNet, src/pyroundtrip/model.py:17-25).
 *   which     RT_MLP_G or RT_MLP_H
 *   n_layers  number of dense layers (hidden layers + the linear output layer), 2..RT_MAX_LAYERS
 *   dims      n_layers+1 ints: dims[0] = input dim, dims[l+1] = width of layer l's output
 *   W[l]      dims[l] x dims[l+1] floats, row-major;  b[l]: dims[l+1] floats
 *   lrelu_alpha  leaky-ReLU slope applied after every layer but the last (0.2 in the reference,
 *             model.py:5-6); must be in [0, 1].
 * Limits: hidden widths <= 1024 (any mix: they are zero-padded, exactly, to 64/128/256/512/1024), dims[0] <= 128,
 * dims[n_layers] <= 128, and a pass must fit the 160 KiB of LDS: 256 B per input dim (rounded up to 8 dims) + the
 * larger of 256 B per unit of the padded hidden width (128 at width 1024) and 1 KiB per output dim (512 B at width 1024)
 * + about 6 KiB — e.g. at width 512 at most 104 input dims, at output dim 128 at most 104 input dims at any width;
 * otherwise RT_EUNSUPPORTED, with the numbers in the message.
 * Nets of width <= 256 with dims[n_layers] <= 32 run as one- or two-wave workgroups (four / two to a CU)
 * with bit-identical results; RT_IS_NARROW=0 in the environment (read here) keeps them on the four-wave kernels.
 */
int rt_set_mlp(rt_ctx* ctx, int which, int n_layers, const int32_t* dims,
               const float* const* W, const float* const* b, float lrelu_alpha);

/* out[n,:] = MLP_which(in[n,:]) for n < N, float32.  Replaces predict_y / predict_x
 * (main_density_est.py:222-234 / 237-249; v2: self.g_net(...) / self.e_net(...), roundtrip.py:249,283). */
int rt_mlp_forward(rt_ctx* ctx, int which, const float* in, int64_t N, float* out);

/* Jacobian of MLP_which at every row of `in`, by forward-mode differentiation on the GPU.  Replaces get_jacobian
 * (main_density_est.py:252-263: sess.run(self.J), with self.J = batch_jacobian(self.y_, self.x), :65; v2
 * tf.GradientTape.batch_jacobian, src/pyroundtrip/roundtrip.py:312-324), which the Laplace estimator
 * estimate_py_with_CF (:351-387) calls at x = H(y).
 *   in   N x dims[0] float32        out  N x dims[L] float32, MLP_which(in) (may be NULL)
 *   J    N x dims[L] x dims[0] float32, J[n][i][j] = d out[n][i] / d in[n][j]  (the reference's (N, y_dim, x_dim))
 * leaky_relu'(t) is taken as 1 for t > 0 and lrelu_alpha otherwise.  Hidden width <= 512 only (RT_EUNSUPPORTED
 * above). */
int rt_mlp_jacobian(rt_ctx* ctx, int which, const float* in, int64_t N, float* out, float* J);

/* xhat = H(y): rt_mlp_forward(ctx, RT_MLP_H, ...) under the name the estimator uses
 * (main_density_est.py:320). */
int rt_encode(rt_ctx* ctx, const float* y, int64_t N, float* xhat);

/* Laplace-approximation (closed-form) estimate of p(y[n]) for n < N: replaces estimate_py_with_CF
 * (main_density_est.py:351-387; v2 estimate_px_with_CF, src/pyroundtrip/roundtrip.py:327-369) as a whole.  Needs G and H.
 *   y   N x dy float64 query points        out  N float64:  log_flag 1: log p(y), 0: p(y)
 * On the device, per query point: xhat = H((float)y) (:363), G(xhat) (:364) and the Jacobian J of G at xhat (:367,
 * rt_mlp_jacobian's kernel), then in float64 (:365-384)
 *   r = y - G(xhat);  A = J^T J;  b = J^T r;  Sigma_inv = I + A / sd_y^2;  mu = inv(Sigma_inv) (b / sd_y^2 - xhat);
 *   c = |xhat|^2 + |r|^2 / sd_y^2 - mu^T Sigma_inv mu;
 *   log p = -dy log(sqrt(2 pi) sd_y) + 0.5 log det inv(Sigma_inv) - 0.5 c                          (:382)
 *   p     = 1 / (sqrt(2 pi) sd_y)^dy * sd_y^dy * sqrt(det inv(Sigma_inv)) * exp(-0.5 c)            (:384)
 * (:384 carries an sd_y^dy factor :382 does not; both are reproduced as written).  The reference inverts Sigma_inv with
 * LAPACK's LU and takes the determinant of the inverse; here the symmetric positive definite Sigma_inv is factored once
 * (Cholesky), which gives the determinant and mu^T Sigma_inv mu = |L^-1 (b / sd_y^2 - xhat)|^2 directly: same value, agreeing
 * with the reference's float64 formulas to ~1e-12 relative on well-conditioned points.  The Jacobians stay on the device.
 * Limits: rt_mlp_jacobian's (hidden width <= 512), and dx*dx + dy + dx float64 values must fit the LDS (dx <= 128 does).
 * rt_cf_estimate_dev: y_dev / out_dev are device pointers; asynchronous on the context's stream (rt_sync). */
int rt_cf_estimate(rt_ctx* ctx, const double* y, int64_t N, double sd_y, int log_flag, double* out);
int rt_cf_estimate_dev(rt_ctx* ctx, const double* y_dev, int64_t N, double sd_y, int log_flag, double* out_dev);

/* Mean and standard error of N float64 values on the device: the validation / test bookkeeping of train()
 * (main_density_est.py:187-188 and :194-195: np.mean(py_est), np.std(py_est) / np.sqrt(len(py_est))).  The sums run in
 * numpy's own order (np.add.reduce: buffers of 8192 elements, pairwise summation with eight partial sums per block of
 * <= 128), the deviations are squared without fused multiply-add, division and square root are correctly rounded: the
 * two results have the bits numpy returns for the same array.  N = 0 gives NaN, NaN (as numpy does).
 *   rt_mean_stderr       values on the host (they are uploaded);  rt_mean_stderr_dev  values and out2 (mean, stderr) on
 *                        the device, asynchronous on the context's stream
 *   rt_is_estimate_summary   rt_is_estimate with log_flag = 1 plus the statistic of its N results, taken where they are
 *                        (no second trip over the bus): what train() does with X_val / X_test at sample_size 2000. */
int rt_mean_stderr(rt_ctx* ctx, const double* values, int64_t N, double* mean, double* stderr_out);
int rt_mean_stderr_dev(rt_ctx* ctx, const double* values_dev, int64_t N, double* out2_dev);
int rt_is_estimate_summary(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base,
                           int64_t S, double sd_y, double scale, double df, uint64_t seed, double* out,
                           double* mean, double* stderr_out);

/* Importance-sampling estimate of p(y[n]) for n < N  (main_density_est.py:321-345).
 *   y      N x dy float64 query points          xhat   N x dx float32 proposal centres H(y)
 *   y_index_base  global index of y[0]: sample i of query point n is drawn from the counter-based
 *          generator at (seed, y_index_base + n, i), so a sharded evaluation is bit-identical to an
 *          unsharded one
 *   S      sample_size       sd_y, scale, df   as in the reference (df: main_density_est.py:41)
 *   log_flag  1: out = log p(y)   0: out = p(y)  (reference `log=`; :339-345)
 *   out    N float64
 * Proposal (replaces sample_from_qx, :310-318): x_i = xhat + scale * z / sqrt(chi2_df / df), z ~ N(0,I).
 * Score (replaces py_given_x :278-290 and w_likelihood_ratio :292-307), float64:
 *   v_i = -dy*log(sqrt(2pi)*sd_y) - |y - G(x_i)|^2 / (2 sd_y^2) - dx*log(sqrt(2pi)) - |x_i|^2/2
 *         - sum_d t.logpdf(xhat_d - x_id, df, 0, scale)
 * Reduction (:339-345): log( sum_i exp(v_i) ) - log(S).
 * Non-finite scores behave as in the reference (:340-345): a score of -inf (G overflowed to inf) weighs 0; a NaN or +inf
 * score makes the point's estimate NaN; if every score is -inf the estimate is NaN under log_flag = 1 ((-inf) - (-inf))
 * and 0 under log_flag = 0.
 * A call of more than 2^28 sample-evals (N*S; the environment variable RT_IS_MAX_LAUNCH_EVALS, read by rt_ctx_create,
 * overrides the bound) runs as several kernel launches over blocks of query points; the result is the same.
 */
int rt_is_estimate(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base,
                   int64_t S, double sd_y, double scale, double df, uint64_t seed, int log_flag,
                   double* out);

/* The same estimate for n_sd (<= RT_MAX_SD) values of sd_y at once: out is n_sd x N (row k = sd_y[k]).
 * Replaces the inner loop of model_selection (main_density_est.py:207-219), which calls the estimator once
 * per (sd, scale) pair after reseeding with 0 each time: for a fixed scale the draws — and therefore G(x_i) —
 * are the same for every sd, so one sampler + G pass serves all of them (3x fewer flops for its sd_list).
 * Row k is bit-identical to rt_is_estimate(..., sd_y[k], ...). */
int rt_is_estimate_multi(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base,
                         int64_t S, int n_sd, const double* sd_y, double scale, double df, uint64_t seed,
                         int log_flag, double* out);

/* Same, with the proposal samples supplied by the caller instead of drawn on the GPU (parity mode):
 *   x_samples  N x S x dx float64 — what sample_from_qx (:310-318) returned for each query point.
 * G sees (float)x (the reference feeds a float32 placeholder, :232), the score uses the float64 x. */
int rt_is_estimate_given_samples(rt_ctx* ctx, const double* y, const float* xhat,
                                 const double* x_samples, int64_t N, int64_t S, double sd_y,
                                 double scale, double df, int log_flag, double* out);

/* Writes the float32 proposal samples rt_is_estimate would use: x_samples is N x S x dx.
 * (Exposes sample_from_qx, :310-318, for tests.) */
int rt_draw_samples(rt_ctx* ctx, const float* xhat, int64_t N, int64_t y_index_base, int64_t S,
                    double scale, double df, uint64_t seed, float* x_samples);

/* The sampler's counter-based generator on caller-chosen words, so that it can be pinned to the published algorithm
 * (Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; known-answer
 * vectors of the Random123 distribution): ctr_key is n x 6 uint32 (counter words 0..3, key words 0..1), out n x 4.
 * rt_draw_samples calls it with counter = (sample index, call index, y index lo, y index hi), key = seed (lo, hi).
 * Replaces numpy's MT19937 behind np.random.chisquare / multivariate_normal (main_density_est.py:267,316-317). */
int rt_philox4x32_10(rt_ctx* ctx, const uint32_t* ctr_key, int64_t n, uint32_t* out);

/* Diagnostic: out[i] = log(1 + a[i]) for a[i] >= 0 (+inf -> +inf, NaN -> NaN) as the fused kernels' scorer evaluates the
 * Student-t proposal's log1p(u^2 / df) (scipy.stats.t.logpdf inside w_likelihood_ratio,
 * /root/reference/main_density_est.py:301-305): the device function itself (even i: inlined; odd i: behind a call, the two
 * forms the kernels use), for accuracy tests against libm.  Host arrays; synchronous. */
int rt_log1p_pos(rt_ctx* ctx, const double* a, int64_t n, double* out);

/* Device-resident variant of rt_is_estimate: y_dev, xhat_dev, out_dev are DEVICE pointers on the
 * context's device.  Enqueues on the context's stream and returns without waiting; call rt_sync()
 * before reading out_dev from another stream or the host. */
int rt_is_estimate_dev(rt_ctx* ctx, const double* y_dev, const float* xhat_dev, int64_t N,
                       int64_t y_index_base, int64_t S, double sd_y, double scale, double df,
                       uint64_t seed, int log_flag, double* out_dev);

/* rt_is_estimate_dev for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc.
 * NO COUNTERPART IN THE REFERENCE.  Same arguments, same kernels, bit-identical results; it calls no hipMalloc / hipFree /
 * hipEventCreate (so rt_get_counters does not see its launches).  The one buffer the estimate needs beyond its inputs and
 * outputs — the per-pass (max v, sum exp(v - max)) pairs, 16 bytes per pass of every query point — lives in a file-scope
 * __device__ array of RT_NOALLOC_SCRATCH_BYTES (32 MiB) that every GPU the library runs on holds until the process
 * ends.  A call whose pairs do not fit,
 *     N * E * 16  >  RT_NOALLOC_SCRATCH_BYTES,   E = rt_is_pair_count(S)   (E = ceil(S / 64) unless the last S mod 64 <= 32
 *                                                 samples are packed, 32-sample passes at hidden width 1024;
 *                                                 e.g. N > 13 357 at S = 10 000, width 512),
 * is refused with RT_ETOOBIG and a message that names the largest N for this S: the bound is on the WHOLE call's N
 * (although its launches reuse the array block by block); split larger calls.  Every pair a call reads it has written
 * first, so the array needs no clearing between calls.  All contexts on one GPU share the array: calls of one context are
 * ordered by its stream; a call made while ANOTHER context's call on the same GPU is still outstanding is refused with
 * RT_EBUSY (serialise the callers or rt_sync the other context first) rather than left to corrupt both results.
 * (Not for HIP graph capture: the launches go to the context's own stream, and the launch sequence — the XCD-share
 * update, a process's share-measuring first block — is decided at enqueue time.) */
int rt_is_estimate_dev_noalloc(rt_ctx* ctx, const double* y_dev, const float* xhat_dev, int64_t N,
                               int64_t y_index_base, int64_t S, double sd_y, double scale, double df,
                               uint64_t seed, int log_flag, double* out_dev);

/* --- Splitting one query point's SAMPLES over callers (second axis: fewer query points than GPUs) ------
 * The fused kernel reduces every pass (64 samples; 32 at hidden width 1024) of a query point to a pair
 * (max v, sum exp(v - max)); in packed mode the last S mod 64 samples of a point form one more pair.  rt_is_estimate
 * merges a point's E pairs in a fixed order.  These three calls expose that: ranks compute disjoint ranges of the E
 * pairs, exchange them (16 bytes per pair and sd value), and merge — the result is bit-identical to rt_is_estimate's.
 * No counterpart in the reference (its loop over samples, main_density_est.py:331-335, is not distributed). */

/* E = number of pairs per query point and sd value for sample_size S under the G that is set. */
int rt_is_pair_count(rt_ctx* ctx, int64_t S, int64_t* E);

/* Pairs [e_lo, e_hi) of every query point: pairs is N x (e_hi - e_lo) x n_sd x 2 float64 (max, sum).  Arguments as
 * rt_is_estimate_multi. */
int rt_is_pairs(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base, int64_t S,
                int n_sd, const double* sd_y, double scale, double df, uint64_t seed, int64_t e_lo, int64_t e_hi,
                double* pairs);

/* out (n_sd x N) = the estimates from all E pairs of every point: pairs is N x E x n_sd x 2, S the sample_size they
 * were computed for (the reduction of main_density_est.py:339-345: log(sum exp v) - log S, or its exp). */
int rt_is_merge_pairs(rt_ctx* ctx, const double* pairs, int64_t N, int64_t E, int n_sd, int64_t S, int log_flag,
                      double* out);

/* Device-resident rt_mlp_forward (in_dev/out_dev are device pointers); asynchronous like the above. */
int rt_mlp_forward_dev(rt_ctx* ctx, int which, const float* in_dev, int64_t N, float* out_dev);

/* Waits for everything enqueued on the context's stream. */
int rt_sync(rt_ctx* ctx);

/* Kernel-time / flop counters accumulated since the last rt_reset_counters (waits for the stream).  Every launch of
 * the fused kernel is bracketed by a pair of HIP events; a context keeps at most 512 pairs and folds them into the
 * running sum (waiting for the newest one) when that many launches are outstanding, so the asynchronous entry points
 * block briefly once per 512 launches even if the counters are never read. */
int rt_get_counters(rt_ctx* ctx, rt_counters* out);
int rt_reset_counters(rt_ctx* ctx);

/* The instantiation of the fused kernel rt_is_estimate launches for the G that is set and this df (hidden width the net
 * is padded to, sample tiles per wave, given-samples flag, shared-sampler flag, waves per workgroup), e.g.
 * "is_fused_kernel<512, 2, false, false, 4>": the name a rocprofv3 kernel trace shows, for reports (bench.py). */
int rt_is_kernel_name(rt_ctx* ctx, double df, char* buf, size_t cap);

/* Device facts the host side needs for work partitioning. */
int rt_device_cu_count(rt_ctx* ctx, int* out);

/* The shares of the fused kernel's passes that the eight workgroup classes blockIdx % 8 — the eight XCDs, to which
 * workgroups are dispatched round-robin — get in this context's next chip-filling launch (they sum to 1; 1/8 each at
 * first).  The XCDs of one MI355X do not run at one clock (1-2 % apart, stable from launch to launch), and a persistent
 * kernel with equal shares ends with its slowest XCD; after every long launch the library re-derives the shares, on the
 * device, from how long each class took; the first long call of a process on a GPU runs a short first block of its
 * query points (one more launch) to measure them for the rest of the call.  Results never depend on them.
 * RT_IS_BALANCE=0 (read by rt_ctx_create) keeps equal shares.  No counterpart in the reference.  Waits for the context's
 * stream. */
int rt_get_xcd_shares(rt_ctx* ctx, double* shares8);

/* CRC-32C (Castagnoli) of n bytes, continuing from `crc` (0 to start): the checksum TensorFlow's checkpoint files carry
 * per index block and per tensor, which `tf.train.Saver.restore` / `ckpt.restore` verify (main_density_est.py:402,410;
 * roundtrip.py:57) and roundtrip_amd/checkpoint.py verifies through this.  Host code, needs no context or GPU. */
uint32_t rt_crc32c(const void* data, size_t n, uint32_t crc);

/* --- Device memory for callers of the *_dev entry points (the reference keeps its arrays in numpy and feeds TF
 * placeholders, main_density_est.py:232,247; a host program that wants its query points resident in HBM needs no other
 * GPU library than this one).  rt_dev_free waits for the context's stream first; the copies are synchronous. */
int rt_dev_alloc(rt_ctx* ctx, size_t bytes, void** out_dev);
int rt_dev_free(rt_ctx* ctx, void* ptr_dev);
/* Free and total device memory in bytes (hipMemGetInfo) on the context's device: lets a caller — and this repo's test of
 * rt_is_estimate_dev_noalloc — see that a call allocated nothing. */
int rt_dev_mem_info(rt_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int rt_memcpy_h2d(rt_ctx* ctx, void* dst_dev, const void* src, size_t bytes);
int rt_memcpy_d2h(rt_ctx* ctx, void* dst, const void* src_dev, size_t bytes);

/* --- The final gather over GPUs ----------------------------------------------------------------------
 * Query points shard over ranks with no data-path collective; the one exchange is the all-gather that assembles the
 * N results (replaces nothing: the reference's loop over points, main_density_est.py:324, is serial; the nearest
 * analogue is the `np.array([...])` that collects the per-point values at :342).  It runs on RCCL, on the context's
 * stream; librccl.so.1 is dlopen'ed by the first rt_comm_* call (RT_RCCL_LIB overrides the name), so one-GPU processes
 * never load it.  One communicator per context; one rank per GPU.
 *   rt_comm_unique_id   rank 0 makes the RT_COMM_ID_BYTES rendezvous token and hands it to the other ranks by any
 *                       means (roundtrip_amd/dist.py: a TCP exchange on MASTER_ADDR)
 *   rt_comm_init        collective over all nranks ranks
 *   rt_comm_all_gather_f64      recv[r*count + i] = rank r's send[i]; host buffers, synchronous
 *   rt_comm_all_gather_f64_dev  the same on device buffers, asynchronous on the context's stream (after the estimate
 *                       enqueued there: no host round trip between kernel and gather)
 *   rt_comm_all_reduce_max_f64  in place, host buffer (bench.py's max-over-ranks time); rt_comm_barrier = one of them */
int rt_comm_unique_id(void* id128);
int rt_comm_init(rt_ctx* ctx, int nranks, int rank, const void* id128);
int rt_comm_destroy(rt_ctx* ctx);
int rt_comm_info(rt_ctx* ctx, int* rank, int* nranks);
int rt_comm_all_gather_f64(rt_ctx* ctx, const double* send, int64_t count, double* recv);
int rt_comm_all_gather_f64_dev(rt_ctx* ctx, const double* send_dev, int64_t count, double* recv_dev);
int rt_comm_all_reduce_max_f64(rt_ctx* ctx, double* values, int64_t count);
int rt_comm_barrier(rt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RT_IS_H */
