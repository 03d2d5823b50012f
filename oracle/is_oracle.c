/*
 * is_oracle.c — CPU restatement of Roundtrip's importance-sampling density estimator.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from it.
 * Nothing under roundtrip_amd/ imports, links or executes it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden vectors or fixtures for this
 * path and cannot be run here (TensorFlow is not installed), so this restatement is
 * pinned only by (a) the numpy/scipy restatement oracle/is_ref.py, which calls the reference's own
 * third-party arithmetic (numpy legacy RandomState, scipy.stats.t.logpdf) statement by statement, and
 * (b) closed-form known answers (tests/test_oracle.py).
 *
 * What it follows (paths relative to /root/reference):
 *   rto_mlp_forward               model.py:61-88 (Generator.__call__: fully_connected = matmul + bias,
 *                                 leaky_relu model.py:5-6), fed float32 as main_density_est.py:232,247;
 *                                 v2 twin src/pyroundtrip/model.py:29-50
 *   score_point                   py_given_x main_density_est.py:278-290, w_likelihood_ratio :292-307,
 *                                 reduction :339-345 (v2: roundtrip.py:241-270, 303-309)
 *   rto_mlp_jacobian              get_jacobian main_density_est.py:252-263 (self.J, :65), forward mode
 *   rto_draw_samples              NOT the reference's sampler (numpy's global MT19937 stream, :316-318,
 *                                 lives in is_ref.py): it restates this repo's Philox4x32-10 proposal
 *                                 sampler so that tests can check the GPU draws bit-for-bit-ish.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_MAX_LAYERS 24

typedef struct {
    int n_layers;
    int dims[RTO_MAX_LAYERS + 1];
    float* W[RTO_MAX_LAYERS]; /* [in][out] row-major */
    float* b[RTO_MAX_LAYERS];
    float alpha;
} rto_mlp;

void* rto_mlp_create(int n_layers, const int32_t* dims, const float* const* W, const float* const* b,
                     float alpha) {
    if (n_layers < 1 || n_layers > RTO_MAX_LAYERS) return NULL;
    rto_mlp* m = (rto_mlp*)calloc(1, sizeof(rto_mlp));
    if (!m) return NULL;
    m->n_layers = n_layers;
    m->alpha = alpha;
    for (int l = 0; l <= n_layers; ++l) m->dims[l] = dims[l];
    for (int l = 0; l < n_layers; ++l) {
        size_t nw = (size_t)dims[l] * dims[l + 1];
        m->W[l] = (float*)malloc(nw * sizeof(float));
        m->b[l] = (float*)malloc((size_t)dims[l + 1] * sizeof(float));
        memcpy(m->W[l], W[l], nw * sizeof(float));
        memcpy(m->b[l], b[l], (size_t)dims[l + 1] * sizeof(float));
    }
    return m;
}

void rto_mlp_destroy(void* p) {
    rto_mlp* m = (rto_mlp*)p;
    if (!m) return;
    for (int l = 0; l < m->n_layers; ++l) { free(m->W[l]); free(m->b[l]); }
    free(m);
}

/* leaky_relu, model.py:5-6: max(min(0, a*x), x); identical to max(x, a*x) for 0 <= a <= 1. */
static inline float lrelu(float x, float a) { return fmaxf(fminf(0.0f, a * x), x); }

/*
 * One dense layer for one row.  acc[f] accumulates in[k]*W[k][f] as a float32 fmaf chain over k.
 *   order 0: k ascending (the plain "x @ W").
 *   order 1: the summation order of the HIP kernel (rt_is.hip): k-groups of 8; inside a group
 *            k = 8g+e, 8g+4+e for e = 0..3 (the two lane halves of v_mfma_f32_32x32x2_f32); the output
 *            layer is split into four K-quarters whose partial sums are added ((p0+p1)+p2)+p3.
 * Bias is added after the matmul (fully_connected = matmul then bias_add), then leaky_relu on hidden layers.
 */
#define RTO_RB 32 /* rows per block: weights are re-used from cache across the block */

/* k visiting order of one dense layer.  Returns the number of k's written to seq; qstart[q] is where
 * K-quarter q begins (order 1, output layer only; otherwise one segment). */
static int k_sequence(const rto_mlp* m, int l, int order, int* seq, int* qstart, int* nq_out) {
    const int K = m->dims[l];
    const int last = (l == m->n_layers - 1);
    int n = 0;
    if (order == 0) {
        for (int k = 0; k < K; ++k) seq[n++] = k;
        qstart[0] = 0; qstart[1] = n; *nq_out = 1;
        return n;
    }
    /* the kernel runs at the hidden width padded up to 128/256/512/1024: its K-quarters are quarters of THAT */
    int hmax = 0;
    for (int i = 1; i < m->n_layers; ++i) if (m->dims[i] > hmax) hmax = m->dims[i];
    const int hid = hmax <= 128 ? 128 : hmax <= 256 ? 256 : hmax <= 512 ? 512 : 1024;
    const int ngrp = (K + 7) / 8;
    const int nq = (last && m->n_layers > 1) ? 4 : 1;
    const int gq = (nq == 4) ? hid / 32 : ngrp; /* groups per quarter */
    /* output layers of width <= 4 run on 4x4x1 MFMAs (rt_is.hip last_layer_small; 64-sample passes, i.e. hidden
     * width <= 512): k ascends inside a quarter */
    const int small = last && nq == 4 && m->dims[l + 1] <= 4 && hid <= 512;
    for (int q = 0; q < nq; ++q) {
        qstart[q] = n;
        for (int g = q * gq; g < (q + 1) * gq && g < ngrp; ++g) {
            if (small) {
                for (int k = 8 * g; k < 8 * g + 8; ++k) if (k < K) seq[n++] = k;
                continue;
            }
            for (int e = 0; e < 4; ++e)
                for (int hh = 0; hh < 2; ++hh) {
                    const int k = 8 * g + 4 * hh + e;
                    if (k < K) seq[n++] = k; /* zero padding: fmaf(0, 0, acc) == acc */
                }
        }
    }
    qstart[nq] = n;
    *nq_out = nq;
    return n;
}

/*
 * One dense layer for a block of R <= RTO_RB rows.  out[r][f] accumulates in[r][k]*W[k][f] as a float32
 * fmaf chain over k, one chain per (r, f):
 *   order 0: k ascending (the plain "x @ W").
 *   order 1: the summation order of the HIP kernel (rt_is.hip): k-groups of 8; inside a group
 *            k = 8g+e, 8g+4+e for e = 0..3 (the two lane halves of v_mfma_f32_32x32x2_f32); the output
 *            layer is split into four K-quarters whose partial sums are added ((p0+p1)+p2)+p3.
 * Bias is added after the matmul (fully_connected = matmul then bias_add), then leaky_relu on hidden layers;
 * order 1 starts a hidden layer's chain FROM the bias instead (what the kernel does, see below).
 * in: [R][ldi], out: [R][ldo], tmp: [R][ldo].
 */
static void dense_block(const rto_mlp* m, int l, int R, const float* in, int ldi, float* out, int ldo,
                        float* tmp, int order, int* seq) {
    const int NF = m->dims[l + 1];
    const float* W = m->W[l];
    const int last = (l == m->n_layers - 1);
    int qstart[5], nq;
    k_sequence(m, l, order, seq, qstart, &nq);
    for (int q = 0; q < nq; ++q) {
        float* dst = (q == 0) ? out : tmp;
        /* order 1, hidden layers: the kernel's first MFMA of a layer takes the bias as its C operand, so the
         * chain starts from the bias instead of adding it at the end */
        const int bias_first = (order == 1 && !last);
        for (int r = 0; r < R; ++r)
            for (int f = 0; f < NF; ++f) dst[r * ldo + f] = bias_first ? m->b[l][f] : 0.0f;
        for (int i = qstart[q]; i < qstart[q + 1]; ++i) {
            const int k = seq[i];
            const float* w = W + (size_t)k * NF;
            for (int r = 0; r < R; ++r) {
                const float a = in[r * ldi + k];
                float* d = dst + r * ldo;
                for (int f = 0; f < NF; ++f) d[f] = fmaf(a, w[f], d[f]);
            }
        }
        if (q > 0)
            for (int r = 0; r < R; ++r)
                for (int f = 0; f < NF; ++f) out[r * ldo + f] = out[r * ldo + f] + tmp[r * ldo + f];
    }
    const float* b = m->b[l];
    for (int r = 0; r < R; ++r) {
        float* o = out + r * ldo;
        if (last) for (int f = 0; f < NF; ++f) o[f] = o[f] + b[f];
        else if (order == 1) for (int f = 0; f < NF; ++f) o[f] = lrelu(o[f], m->alpha);
        else for (int f = 0; f < NF; ++f) o[f] = lrelu(o[f] + b[f], m->alpha);
    }
}

static int max_width(const rto_mlp* m) {
    int w = 0;
    for (int l = 0; l <= m->n_layers; ++l) if (m->dims[l] > w) w = m->dims[l];
    return w;
}

/* scratch for mlp_block: 3 * RTO_RB * maxw floats + maxw ints */
static size_t mlp_scratch_bytes(const rto_mlp* m) {
    const size_t mw = (size_t)max_width(m);
    return 3 * RTO_RB * mw * sizeof(float) + mw * sizeof(int);
}

/* in: [R][dims[0]] float32 -> out: [R][dims[L]] float32, R <= RTO_RB */
static void mlp_block(const rto_mlp* m, int R, const float* in, float* out, void* scratch, int order) {
    const int mw = max_width(m);
    float* a = (float*)scratch;
    float* b = a + (size_t)RTO_RB * mw;
    float* tmp = b + (size_t)RTO_RB * mw;
    int* seq = (int*)(tmp + (size_t)RTO_RB * mw);
    for (int r = 0; r < R; ++r) memcpy(a + (size_t)r * mw, in + (size_t)r * m->dims[0], sizeof(float) * m->dims[0]);
    for (int l = 0; l < m->n_layers; ++l) {
        dense_block(m, l, R, a, mw, b, mw, tmp, order, seq);
        float* t = a; a = b; b = t;
    }
    const int od = m->dims[m->n_layers];
    for (int r = 0; r < R; ++r) memcpy(out + (size_t)r * od, a + (size_t)r * mw, sizeof(float) * od);
}

/* predict_y / predict_x (main_density_est.py:222-249): float32 in, float32 out. */
void rto_mlp_forward(void* p, const float* in, int64_t N, float* out, int order, int nthreads) {
    const rto_mlp* m = (const rto_mlp*)p;
    const int64_t nblk = (N + RTO_RB - 1) / RTO_RB;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        void* scratch = malloc(mlp_scratch_bytes(m));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t bk = 0; bk < nblk; ++bk) {
            const int64_t r0 = bk * RTO_RB;
            const int R = (int)((N - r0 < RTO_RB) ? N - r0 : RTO_RB);
            mlp_block(m, R, in + r0 * m->dims[0], out + r0 * m->dims[m->n_layers], scratch, order);
        }
        free(scratch);
    }
}

/*
 * get_jacobian (main_density_est.py:252-263: sess.run(self.J), self.J = batch_jacobian(self.y_, self.x) :65; v2
 * tf.GradientTape.batch_jacobian, src/pyroundtrip/roundtrip.py:312-324), restated as forward-mode differentiation
 * of the same float32 MLP: for input direction d the tangent of a dense layer is the same fmaf chain over k
 * without the bias, and leaky_relu'(t) is 1 where t > 0 and alpha elsewhere (TensorFlow's reverse mode computes
 * the same products in another order; at exactly t == 0 TF's maximum/minimum gradient convention gives 0, a
 * measure-zero difference).  out: N x dims[L] (may be NULL), J: N x dims[L] x dims[0], row-major.
 * order as in dense_block (order 1 = the HIP kernel's summation order: bit-exact comparisons).
 */
void rto_mlp_jacobian(void* p, const float* in, int64_t N, float* out, float* J, int order, int nthreads) {
    const rto_mlp* m = (const rto_mlp*)p;
    const int L = m->n_layers, id = m->dims[0], od = m->dims[L];
    const int mw = max_width(m);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* a = (float*)malloc(sizeof(float) * 4 * (size_t)mw);
        float *ta = a + mw, *b = a + 2 * mw, *tb = a + 3 * mw;
        int* seq = (int*)malloc(sizeof(int) * (size_t)mw);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t n = 0; n < N; ++n)
            for (int d = 0; d < id; ++d) {
                float *pa = a, *pta = ta, *pb = b, *ptb = tb;
                for (int k = 0; k < id; ++k) { pa[k] = in[n * id + k]; pta[k] = (k == d) ? 1.0f : 0.0f; }
                for (int l = 0; l < L; ++l) {
                    const int NF = m->dims[l + 1], last = (l == L - 1);
                    const float* W = m->W[l];
                    int qstart[5], nq;
                    k_sequence(m, l, order, seq, qstart, &nq);
                    const int bias_first = (order == 1 && !last);
                    for (int f = 0; f < NF; ++f) {
                        float ps = 0.0f, ts = 0.0f;
                        for (int q = 0; q < nq; ++q) {
                            float cp = (bias_first && q == 0) ? m->b[l][f] : 0.0f, ct = 0.0f;
                            for (int i = qstart[q]; i < qstart[q + 1]; ++i) {
                                const int k = seq[i];
                                cp = fmaf(pa[k], W[(size_t)k * NF + f], cp);
                                ct = fmaf(pta[k], W[(size_t)k * NF + f], ct);
                            }
                            if (q == 0) { ps = cp; ts = ct; } else { ps = ps + cp; ts = ts + ct; }
                        }
                        if (last) { pb[f] = ps + m->b[l][f]; ptb[f] = ts; }
                        else {
                            const float t = bias_first ? ps : ps + m->b[l][f];
                            pb[f] = lrelu(t, m->alpha);
                            ptb[f] = (t > 0.0f) ? ts : ts * m->alpha;
                        }
                    }
                    float* x;
                    x = pa; pa = pb; pb = x;
                    x = pta; pta = ptb; ptb = x;
                }
                if (d == 0 && out) for (int f = 0; f < od; ++f) out[n * od + f] = pa[f];
                for (int f = 0; f < od; ++f) J[((size_t)n * od + f) * id + d] = pta[f];
            }
        free(a);
        free(seq);
    }
}

/* scipy.stats.t.logpdf(x, df, loc=0, scale) (main_density_est.py:301):
 * lgamma((df+1)/2) - lgamma(df/2) - 0.5*log(df*pi) - (df+1)/2 * log1p(u^2/df) - log(scale), u = x/scale */
double rto_t_logpdf(double x, double df, double scale) {
    const double u = x / scale;
    return lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * M_PI) -
           0.5 * (df + 1.0) * log1p(u * u / df) - log(scale);
}

/* One query point given its S proposal samples (float64).  v[] is scratch of S doubles. */
static double score_point(const rto_mlp* g, const double* y, const float* xhat, const double* xs,
                          int64_t S, double sd_y, double scale, double df, int log_flag, int order,
                          double* v, void* scratch, float* xin, float* yhat_blk) {
    const int dx = g->dims[0], dy = g->dims[g->n_layers];
    double vmax = -INFINITY;
    for (int64_t i0 = 0; i0 < S; i0 += RTO_RB) {
        const int R = (int)((S - i0 < RTO_RB) ? S - i0 : RTO_RB);
        for (int r = 0; r < R; ++r)
            for (int d = 0; d < dx; ++d) xin[r * dx + d] = (float)xs[(i0 + r) * dx + d]; /* float32 feed, :232 */
        mlp_block(g, R, xin, yhat_blk, scratch, order);
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r;
            const double* x = xs + i * dx;
            const float* yhat = yhat_blk + r * dy;
            double se = 0.0;
            for (int d = 0; d < dy; ++d) { const double t = y[d] - (double)yhat[d]; se += t * t; }
            double xx = 0.0;
            for (int d = 0; d < dx; ++d) xx += x[d] * x[d];
            if (log_flag) {
                /* :288 */
                const double lpy = -(double)dy * log(sqrt(2.0 * M_PI) * sd_y) - se / (2.0 * sd_y * sd_y);
                /* :301-303 */
                double log_qx = 0.0;
                for (int d = 0; d < dx; ++d) log_qx += rto_t_logpdf((double)xhat[d] - x[d], df, scale);
                const double log_px = -(double)dx * log(sqrt(2.0 * M_PI)) - xx / 2.0;
                v[i] = lpy + (log_px - log_qx);
                if (v[i] > vmax) vmax = v[i];
            } else {
                /* :290, :305-307: linear space, literally */
                const double py = 1.0 / pow(sqrt(2.0 * M_PI) * sd_y, (double)dy) * exp(-se / (2.0 * sd_y * sd_y));
                double qx = 1.0;
                for (int d = 0; d < dx; ++d) qx *= exp(rto_t_logpdf((double)xhat[d] - x[d], df, scale));
                const double px = 1.0 / pow(sqrt(2.0 * M_PI), (double)dx) * exp(-xx / 2.0);
                v[i] = py * (px / qx);
            }
        }
    }
    if (log_flag) {
        /* :340-342: log(sum(exp(v - vmax))) - log(S) + vmax */
        double s = 0.0;
        for (int64_t i = 0; i < S; ++i) s += exp(v[i] - vmax);
        return log(s) - log((double)S) + vmax;
    }
    double s = 0.0; /* :345 np.mean */
    for (int64_t i = 0; i < S; ++i) s += v[i];
    return s / (double)S;
}

/* estimate_py_with_IS with the samples supplied (x_samples: N x S x dx float64). */
int rto_is_estimate_given_samples(void* gp, const double* y, const float* xhat, const double* xs,
                                  int64_t N, int64_t S, double sd_y, double scale, double df,
                                  int log_flag, int order, int nthreads, double* out) {
    const rto_mlp* g = (const rto_mlp*)gp;
    const int dx = g->dims[0], dy = g->dims[g->n_layers];
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double* v = (double*)malloc(sizeof(double) * (size_t)S);
        void* scratch = malloc(mlp_scratch_bytes(g));
        float* xin = (float*)malloc(sizeof(float) * RTO_RB * dx);
        float* yb = (float*)malloc(sizeof(float) * RTO_RB * dy);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t n = 0; n < N; ++n)
            out[n] = score_point(g, y + n * dy, xhat + n * dx, xs + (size_t)n * S * dx, S, sd_y, scale, df,
                                 log_flag, order, v, scratch, xin, yb);
        free(v); free(scratch); free(xin); free(yb);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * This repo's proposal sampler (rt_is.hip draw_sample), restated: Philox4x32-10 keyed by the seed,
 * counter (sample index, call index, y index lo, y index hi); call c yields normals g[4c..4c+3] by two
 * float32 Box-Muller transforms (r = sqrt(-2 ln2 log2 u1), angle u2 revolutions); chi2 = g[0]^2+..+g[ndf-1]^2 for integer df <= 32;
 * x_d = xhat_d + scale*g[ndf+d]/sqrt(chi2/df); any other df: chi2_over_df_gamma below.  libm's logf/sincosf may differ from the GPU's in the
 * last ulp, so agreement with the GPU draws is to a few ulp, not bitwise.
 * --------------------------------------------------------------------------------------------- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
/* Philox4x32-10 on raw words, for the known-answer test against the published Random123 vectors
 * (Salmon et al., SC'11; kat_vectors of the Random123 distribution): out = philox4x32_10(ctr, key). */
void rto_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox(c, key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}
static float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); } /* (k+1/2)/2^23: exact, in (0,1) */
/* rt_sampler.hpp box_muller: r = sqrt((-2 ln 2) log2 u1), the angle u2 in revolutions.  The GPU evaluates log2, sqrt, sin
 * and cos on its transcendental unit; here the angle's sine and cosine are taken in double and rounded once. */
static void box_muller(uint32_t a, uint32_t b, float* n0, float* n1) {
    const float r = sqrtf(-1.38629436111989061883f * log2f(u01(a)));
    const double th = 6.28318530717958647692 * (double)u01(b);
    *n0 = r * (float)cos(th);
    *n1 = r * (float)sin(th);
}

/* chi2(df)/df for a df that is not an integer in [1, 32] (rt_sampler.hpp chi2_over_df_gamma): chi2 = 2*Gamma(df/2, 1)
 * by Marsaglia & Tsang (2000), float32 throughout, on Philox calls 0x40000000, 0x40000001, ... of the sample: shape < 1
 * takes one call for the boost u^(1/shape) and continues with shape + 1; every further call is one proposal
 * (z from words 0 and 1 by Box-Muller, u from word 2), accepted when log u < z^2/2 + d - d v + d log v. */
static float chi2_over_df_gamma(float df, uint32_t i, uint32_t ylo, uint32_t yhi, uint32_t k0, uint32_t k1) {
    float shape = 0.5f * df, boost = 1.0f;
    uint32_t call = 0x40000000u;
    if (shape < 1.0f) {
        uint32_t c[4] = {i, call++, ylo, yhi};
        philox(c, k0, k1);
        boost = powf(u01(c[0]), 1.0f / shape);
        shape += 1.0f;
    }
    const float d = shape - (1.0f / 3.0f), cc = 1.0f / sqrtf(9.0f * d);
    float g = d;
    for (int it = 0; it < 64; ++it) {
        uint32_t c[4] = {i, call++, ylo, yhi};
        philox(c, k0, k1);
        float z, unused;
        box_muller(c[0], c[1], &z, &unused);
        float v = 1.0f + cc * z;
        if (v <= 0.0f) continue;
        v = v * v * v;
        const float u = u01(c[2]);
        if (logf(u) < 0.5f * z * z + d - d * v + d * logf(v)) { g = d * v; break; }
    }
    return 2.0f * g * boost / df;
}

int rto_draw_samples(const float* xhat, int64_t N, int64_t y_index_base, int64_t S, int dx, double scale,
                     double df, uint64_t seed, float* xs) {
    if (!(df > 0.0)) return 1;
    const int ndf = (df == floor(df) && df <= 32.0) ? (int)df : 0;   /* 0: the gamma path */
    const float fscale = (float)scale, fdf = (float)df, inv_df = (float)(1.0 / df);
    for (int64_t n = 0; n < N; ++n)
        for (int64_t i = 0; i < S; ++i) {
            const uint64_t yi = (uint64_t)(y_index_base + n);
            float zsum = 0.0f, mult = 0.0f;
            /* mult = scale * rsq(max(chi2 / df, 1e-30)): the clamp of rt_sampler.hpp draw_sample */
            if (ndf == 0)
                mult = fscale / sqrtf(fmaxf(chi2_over_df_gamma(fdf, (uint32_t)i, (uint32_t)yi, (uint32_t)(yi >> 32),
                                                               (uint32_t)seed, (uint32_t)(seed >> 32)), 1e-30f));
            const int total = ndf + dx;
            for (int c = 0; 4 * c < total; ++c) {
                uint32_t ctr[4] = {(uint32_t)i, (uint32_t)c, (uint32_t)yi, (uint32_t)(yi >> 32)};
                philox(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
                float nn[4];
                box_muller(ctr[0], ctr[1], &nn[0], &nn[1]);
                box_muller(ctr[2], ctr[3], &nn[2], &nn[3]);
                for (int k = 0; k < 4; ++k) {
                    const int g = 4 * c + k;
                    if (g < ndf) {
                        zsum += nn[k] * nn[k];
                        if (g == ndf - 1) mult = fscale / sqrtf(fmaxf(zsum * inv_df, 1e-30f));
                    } else if (g < total) {
                        const int d = g - ndf;
                        xs[((size_t)n * S + i) * dx + d] = xhat[n * dx + d] + mult * nn[k];
                    }
                }
            }
        }
    return 0;
}

int rto_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
