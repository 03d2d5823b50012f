// rt_kernels.hpp — the __global__ kernels: fused importance-sampling estimator, its finalizer, plain MLP forward,
// forward-mode Jacobian, sample dump.  Included by rt_is.hip only.
#pragma once

// ----------------------------------------------------------------------------------------------
// Fused IS kernel.
// ----------------------------------------------------------------------------------------------
struct IsArgs {
    MlpArgs g;
    SamplerArgs sa;
    const double* y;        // N x dy
    const float* xhat;      // N x dx
    const double* xs;       // given samples N x S x dx, or nullptr
    double* partial;        // n_units x n_sd x 2 : (max, sum exp(v - max))
    int64_t N, S, y_index_base;
    int64_t n_units;
    int nchunks;            // chunks per query point
    int64_t passes;         // ceil(S / samples per pass)
    // score constants (fp64)
    int n_sd;               // number of sd_y values scored on the same draws and the same G pass (1..RT_MAX_SD)
    double c_all[RT_MAX_SD];     // -dy*log(sqrt(2pi)*sd) - dx*log(sqrt(2pi)) - dx*ct(df, scale)
    double inv_2sd2[RT_MAX_SD];  // 1 / (2 sd^2)
    double inv_scale;
    double inv_df;
    double half_dfp1;       // (df+1)/2
};

struct Carve {
    int x0, act, scr, lse, yb, xh, total;
};

__host__ __device__ inline Carve carve_for(int hid, int ms, int k0, int in_dim, int out_dim) {
    Carve c;
    int off = 0;
    c.x0 = off; off += k0 * ms * 4;
    int act_bytes = hid * ms * 4;
    const int part_bytes = 4 * out_dim * ms * 4;
    if (part_bytes > act_bytes) act_bytes = part_bytes;
    c.act = off; off += act_bytes;
    c.scr = off; off += 2 * NTHREADS * 8;            // per (phase, sample): squared error, log weight
    c.lse = off; off += 2 * RT_MAX_SD * ms * 8;      // per (sd, sample slot): running max, running sum
    c.yb = off; off += ((out_dim * 8 + 15) / 16) * 16;
    c.xh = off; off += ((in_dim * 4 + 15) / 16) * 16;
    c.total = off;
    return c;
}

// One workgroup per CU, up to 512 registers per wave.  NST = 2: 64 samples per pass (hidden width <= 512).
// NST = 1: 32 samples per pass, used for hidden width 1024, whose 64-sample activation image (256 KB)
// would not fit in LDS.
template <int HID, int NST, bool GIVEN>
__global__ __launch_bounds__(NTHREADS) void is_fused_kernel(const IsArgs a) {
    constexpr int MS = 32 * NST;          // samples per pass
    constexpr int NP = NTHREADS / MS;     // threads per sample in the scoring phase
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MlpArgs& m = a.g;
    const int dx = m.in_dim, dy = m.out_dim;
    const Carve cv = carve_for(HID, MS, m.k0, dx, dy);
    char* x0 = smem + cv.x0;
    char* act = smem + cv.act;
    double* scr = (double*)(smem + cv.scr);
    double* lse = (double*)(smem + cv.lse);
    double* yb = (double*)(smem + cv.yb);
    float* xh = (float*)(smem + cv.xh);
    const float* part = (const float*)act;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int s = tid % MS, p = tid / MS;   // scoring role: sample slot s, dimension phase p

    // zero the padded rows of the x0 image once: rows >= dx are never written again
    for (int i = tid; i < m.k0 * MS / 4; i += NTHREADS) ((f32x4*)x0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef RT_STAMPS
    Stamps st;
    for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
    st.t = __builtin_amdgcn_s_memtime();
#endif
    // contiguous block of work units per workgroup: every workgroup gets the same mix of chunk sizes
    const int64_t u_lo = a.n_units * (int64_t)blockIdx.x / gridDim.x;
    const int64_t u_hi = a.n_units * ((int64_t)blockIdx.x + 1) / gridDim.x;
    for (int64_t unit = u_lo; unit < u_hi; ++unit) {
        const int64_t n = unit / a.nchunks;
        const int c = (int)(unit - n * a.nchunks);
        // chunk c = passes [c*P/nchunks, (c+1)*P/nchunks): sizes differ by at most one pass
        const int64_t s_begin = (a.passes * c / a.nchunks) * MS;
        const int64_t s_hi = (a.passes * (c + 1) / a.nchunks) * MS;
        const int64_t s_end = (s_hi < a.S) ? s_hi : a.S;
        __syncthreads();   // previous unit's readers of yb / xh are done
        for (int d = tid; d < dy; d += NTHREADS) yb[d] = a.y[n * dy + d];
        for (int d = tid; d < dx; d += NTHREADS) xh[d] = a.xhat[n * dx + d];
        __syncthreads();

        // online log-sum-exp state per (sd, sample slot) lives in LDS: lse[(2k)*MS + s] = max, [(2k+1)*MS + s] = sum
        if (tid < MS)
            for (int k = 0; k < a.n_sd; ++k) { lse[(2 * k) * MS + s] = -INFINITY; lse[(2 * k + 1) * MS + s] = 0.0; }

        for (int64_t sb = s_begin; sb < s_end; sb += MS) {
            const int64_t si = sb + s;           // this thread's sample index
            ST(7)
            MlpPre<HID> pre;
            mlp_prefetch<HID, NST>(m, pre, wave, lane);
            // ---- A. proposal samples -> x0 image --------------------------------------------------
            if (GIVEN) {
                if (si < s_end) {
                    const double* src = a.xs + ((size_t)n * a.S + si) * dx;
                    for (int d = p; d < dx; d += NP)
                        *(float*)(x0 + act_off<MS>(d >> 3, s, (d >> 2) & 1) + 4 * (d & 3)) = (float)src[d];
                } else {
                    for (int d = p; d < dx; d += NP)
                        *(float*)(x0 + act_off<MS>(d >> 3, s, (d >> 2) & 1) + 4 * (d & 3)) = 0.0f;
                }
            } else if (tid < MS) {
                draw_sample(a.sa, (uint64_t)(a.y_index_base + n), (uint32_t)si, xh, dx,
                            [&](int d, float v) {
                                *(float*)(x0 + act_off<MS>(d >> 3, s, (d >> 2) & 1) + 4 * (d & 3)) = v;
                            });
            }
            __syncthreads();
            ST(0)
            // ---- B. G forward ---------------------------------------------------------------------
            mlp_core<HID, NST>(m, pre, x0, act, wave, lane ST_ARG);
            // ---- C. score: thread (s, p) handles dims d = p, p+NP, ... -----------------------------
            double se = 0.0;
            for (int d = p; d < dy; d += NP) {
                const double diff = yb[d] - (double)mlp_out<MS>(m, part, d, s);
                se += diff * diff;
            }
            double lw = 0.0;
            for (int d = p; d < dx; d += NP) {
                double x;
                if (GIVEN) x = (si < s_end) ? a.xs[((size_t)n * a.S + si) * dx + d] : 0.0;
                else x = (double)*(const float*)(x0 + act_off<MS>(d >> 3, s, (d >> 2) & 1) + 4 * (d & 3));
                const double u = ((double)xh[d] - x) * a.inv_scale;
                lw += a.half_dfp1 * log1p(u * u * a.inv_df) - 0.5 * x * x;
            }
            scr[p * MS + s] = se;
            scr[NTHREADS + p * MS + s] = lw;
            __syncthreads();
            // ---- D. online log-sum-exp (one thread per sample; one state per sd_y) ------------------
            if (tid < MS && si < s_end) {
                double se_t = scr[s], lw_t = scr[NTHREADS + s];
#pragma unroll
                for (int q = 1; q < NP; ++q) { se_t += scr[q * MS + s]; lw_t += scr[NTHREADS + q * MS + s]; }
                for (int k = 0; k < a.n_sd; ++k) {
                    const double v = a.c_all[k] + (lw_t - se_t * a.inv_2sd2[k]);
                    if (v > -INFINITY && v < INFINITY) {
                        double run_m = lse[(2 * k) * MS + s], run_s = lse[(2 * k + 1) * MS + s];
                        if (v > run_m) { run_s = run_s * exp(run_m - v) + 1.0; run_m = v; }
                        else run_s += exp(v - run_m);
                        lse[(2 * k) * MS + s] = run_m;
                        lse[(2 * k + 1) * MS + s] = run_s;
                    }
                }
            }
        }
        if (wave == 0) {
            for (int k = 0; k < a.n_sd; ++k) {
                const double run_m = (lane < MS) ? lse[(2 * k) * MS + lane] : -INFINITY;
                const double run_s = (lane < MS) ? lse[(2 * k + 1) * MS + lane] : 0.0;
                double M = run_m;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) M = fmax(M, __shfl_xor(M, o));
                double t = (run_m > -INFINITY) ? run_s * exp(run_m - M) : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                if (lane == 0) {
                    a.partial[2 * (unit * a.n_sd + k)] = M;
                    a.partial[2 * (unit * a.n_sd + k) + 1] = t;
                }
            }
        }
    }
#ifdef RT_STAMPS
#if RT_STAMPS == 2
    st.acc[0] = __builtin_amdgcn_s_memtime() - st.t;
#endif
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < NSTAMP; ++i) atomicAdd(&g_stamps[wave * NSTAMP + i], st.acc[i]);
#endif
}

// out[k*out_stride + n] for sd index k: merges the chunks of query point n (n < N, this launch's block).
__global__ void is_finalize_kernel(const double* partial, int64_t N, int nchunks, int n_sd, double logS,
                                   int log_flag, double* out, int64_t out_stride) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < n_sd; ++k) {
        const double* p = partial + 2 * (n * nchunks * n_sd + k);   // chunk stride: 2*n_sd doubles
        const int cs = 2 * n_sd;
        double M = -INFINITY;
        for (int c = 0; c < nchunks; ++c) M = fmax(M, p[c * cs]);
        double t = 0.0;
        for (int c = 0; c < nchunks; ++c)
            if (p[c * cs] > -INFINITY) t += p[c * cs + 1] * exp(p[c * cs] - M);
        const double lp = (M > -INFINITY) ? M + log(t) - logS : -INFINITY;
        out[(int64_t)k * out_stride + n] = log_flag ? lp : exp(lp);
    }
}

// ----------------------------------------------------------------------------------------------
// Plain MLP forward (predict_x / predict_y): 32*NST rows per pass.
// ----------------------------------------------------------------------------------------------
template <int HID, int NST>
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_kernel(const MlpArgs m, const float* in, int64_t N,
                                                           float* out) {
    constexpr int MS = 32 * NST;
    constexpr int NP = NTHREADS / MS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Carve cv = carve_for(HID, MS, m.k0, m.in_dim, m.out_dim);
    char* x0 = smem + cv.x0;
    char* act = smem + cv.act;
    const float* part = (const float*)act;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    for (int i = tid; i < m.k0 * MS / 4; i += NTHREADS) ((f32x4*)x0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t npass = (N + MS - 1) / MS;
    for (int64_t ps = blockIdx.x; ps < npass; ps += gridDim.x) {
        const int sl = tid % MS, ph = tid / MS;   // row slot, dimension phase
        const int64_t row = ps * MS + sl;
        __syncthreads();
        for (int d = ph; d < m.in_dim; d += NP)
            *(float*)(x0 + act_off<MS>(d >> 3, sl, (d >> 2) & 1) + 4 * (d & 3)) =
                (row < N) ? in[row * m.in_dim + d] : 0.0f;
        __syncthreads();
#ifdef RT_STAMPS
        Stamps st; st.t = 0; for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
#endif
        MlpPre<HID> pre;
        mlp_prefetch<HID, NST>(m, pre, wave, lane);
        mlp_core<HID, NST>(m, pre, x0, act, wave, lane ST_ARG);
        if (row < N)
            for (int d = ph; d < m.out_dim; d += NP) out[row * m.out_dim + d] = mlp_out<MS>(m, part, d, sl);
    }
}

// ----------------------------------------------------------------------------------------------
// Jacobian of an MLP by forward-mode differentiation (get_jacobian, main_density_est.py:252-263, the
// tf.gradients graph :65; v2 batch_jacobian, roundtrip.py:312-324).  One pass = 32 rows x one input
// direction d: sample tile 0 carries the rows' primal inputs, sample tile 1 the unit tangent e_d for the same
// rows, so column d of every row's Jacobian comes out of tile 1.  J is N x out_dim x in_dim, row-major.
// The primal is recomputed for every d (in_dim x the work of a plain forward; this path runs once per QUERY
// POINT, not per sample, so it is ~1e-4 of the IS estimate's work).
// ----------------------------------------------------------------------------------------------
template <int HID>
__global__ __launch_bounds__(NTHREADS) void mlp_jvp_kernel(const MlpArgs m, const float* in, int64_t N,
                                                           float* yout, float* J) {
    constexpr int MS = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Carve cv = carve_for(HID, MS, m.k0, m.in_dim, m.out_dim);
    char* x0 = smem + cv.x0;
    char* act = smem + cv.act;
    const float* part = (const float*)act;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    for (int i = tid; i < m.k0 * MS / 4; i += NTHREADS) ((f32x4*)x0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t npass = ((N + 31) / 32) * m.in_dim;
    const int sl = tid & 63, ph = tid >> 6;       // column slot, dimension phase
    for (int64_t ps = blockIdx.x; ps < npass; ps += gridDim.x) {
        const int64_t blk = ps / m.in_dim;
        const int dir = (int)(ps - blk * m.in_dim);
        const int64_t row = blk * 32 + (sl & 31);
        __syncthreads();
        for (int d = ph; d < m.in_dim; d += 4) {
            float v;
            if (sl < 32) v = (row < N) ? in[row * m.in_dim + d] : 0.0f;
            else v = (d == dir) ? 1.0f : 0.0f;
            *(float*)(x0 + act_off<MS>(d >> 3, sl, (d >> 2) & 1) + 4 * (d & 3)) = v;
        }
        __syncthreads();
#ifdef RT_STAMPS
        Stamps st; st.t = 0; for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
#endif
        MlpPre<HID> pre;
        mlp_core<HID, 2, true>(m, pre, x0, act, wave, lane ST_ARG);
        if (row < N) {
            if (sl < 32) {
                if (dir == 0 && yout)
                    for (int d = ph; d < m.out_dim; d += 4) yout[row * m.out_dim + d] = mlp_out<MS>(m, part, d, sl);
            } else {
                for (int d = ph; d < m.out_dim; d += 4)
                    J[(row * m.out_dim + d) * m.in_dim + dir] = mlp_out_nobias<MS>(m, part, d, sl);
            }
        }
    }
}

__global__ void draw_samples_kernel(SamplerArgs sa, const float* xhat, int64_t N, int64_t y_index_base,
                                    int64_t S, int dx, float* xs) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * S) return;
    const int64_t n = t / S, i = t - n * S;
    float* dst = xs + t * dx;
    draw_sample(sa, (uint64_t)(y_index_base + n), (uint32_t)i, xhat + n * dx, dx,
                [&](int d, float v) { dst[d] = v; });
}

