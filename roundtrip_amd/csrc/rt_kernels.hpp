// rt_kernels.hpp — the __global__ kernels: fused importance-sampling estimator, its finalizer, plain MLP forward,
// forward-mode Jacobian, sample dump.  Included by rt_is.hip only.
#pragma once

// ----------------------------------------------------------------------------------------------
// Fused IS kernel.
//
// Work is a linear list of PASSES (one pass = MS samples through all of G's layers):
//   main passes      p in [0, N*ppp):  query point n = p / ppp, pass j = p % ppp, samples [j*MS, min(S, (j+1)*MS));
//   leftover passes  p in [N*ppp, p_total)  (packed mode only: rem = S % MS with 0 < rem <= MS/2):  the last `rem`
//                    samples of q = MS / rem consecutive query points share one pass — slot s carries sample
//                    ppp*MS + s % rem of point g*q + s / rem — instead of one 3/4-empty pass per point.
// Workgroup b runs main passes [p_main*b/G, p_main*(b+1)/G) and then leftover passes [L*b/G, L*(b+1)/G): every workgroup
// gets the same number of passes of either kind (+-1) — or, in a launch that fills the chip, the share of its XCD class
// (IsArgs::bal, balance_update_kernel), split evenly inside the class.
// Every pass reduces ITS samples of ONE query point to (max v, sum exp(v - max)) and stores the pair:
//   partial[((n*E + e)*n_sd + k)*2 ..+1],  e = j for a main pass, e = ppp for the point's share of a leftover pass.
// is_finalize_kernel merges a point's E pairs in a fixed order, so the result does not depend on how passes were dealt to
// workgroups, on N, on the launch split or on the GPU count: it is a function of (weights, y, xhat, S, seed, global index).
// ----------------------------------------------------------------------------------------------
struct IsArgs {
    MlpArgs g;
    SamplerArgs sa;
    const double* y;        // N x dy
    const float* xhat;      // N x dx
    const double* xs;       // given samples N x S x dx, or nullptr
    double* partial;        // N x E x n_sd x 2
    int64_t N, S, y_index_base;
    int64_t ppp;            // main passes per query point in THIS launch: entries [j_lo, j_lo + ppp)
    int64_t j_lo;           // first main pass of a point this launch runs (0 unless the samples are split over ranks)
    int64_t left_s0;        // sample index of a point's first leftover sample (full main passes x MS)
    int64_t p_main;         // N * ppp
    int64_t p_total;        // p_main + number of leftover passes
    int64_t E;              // partial entries per query point: ppp + (rem ? 1 : 0)
    int rem;                // packed mode: samples per point in a leftover pass (S % MS); 0 = no leftover passes
    int q;                  // packed mode: query points per leftover pass (MS / rem)
    // score constants (fp64)
    int n_sd;               // number of sd_y values scored on the same draws and the same G pass (1..RT_MAX_SD)
    double c_all[RT_MAX_SD];     // -dy*log(sqrt(2pi)*sd) - dx*log(sqrt(2pi)) - dx*ct(df, scale)
    double inv_2sd2[RT_MAX_SD];  // 1 / (2 sd^2)
    double inv_scale;
    double inv_df;
    double half_dfp1;       // (df+1)/2
    // XCD balance (see balance_update_kernel): cumulative shares of the eight workgroup classes blockIdx % 8, or nullptr
    // for equal shares; and where every workgroup leaves its start / end time (nullptr: nowhere)
    const double* bal;
    unsigned long long* wg_times;
};

struct Carve {
    int x0, act, scr, yb, xh, ob, total;
};

__host__ __device__ inline Carve carve_for(int hid, int ms, int k0, int in_dim, int out_dim, int nthreads = NTHREADS) {
    Carve c;
    int off = 0;
    c.x0 = off; off += k0 * ms * 4;
    int act_bytes = hid * ms * 4;
    const int part_bytes = 4 * out_dim * ms * 4;
    if (part_bytes > act_bytes) act_bytes = part_bytes;
    c.act = off; off += act_bytes;
    c.scr = off; off += 2 * (nthreads < NTHREADS ? NTHREADS : nthreads) * 8;   // per (phase, sample): squared error, log weight
    c.yb = off; off += ((out_dim * 8 + 15) / 16) * 16;
    c.xh = off; off += ((in_dim * 4 + 15) / 16) * 16;
    c.ob = off; off += ((out_dim * 4 + 15) / 16) * 16;   // the output layer's bias, for the scorer (SHARE and NW != 4 kernels)
    c.total = off;
    return c;
}

// Wave-wide fp64 reductions on DPP moves (an inclusive scan: row_shr 1, 2, 4, 8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3; lane 63 holds the total, read back to every lane).
// About 20 VALU instructions — the ds_bpermute butterfly of __shfl_xor costs ~3000 cycles of dependent LDS latency,
// which matters here because the fused kernel reduces once per pass.  Fixed evaluation order: deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v, double identity) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned long long id = (unsigned long long)__double_as_longlong(identity);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)id, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(id >> 32), (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double read_lane63(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_max(double v) {      // v must not be NaN
    v = fmax(v, dpp_fetch<0x111, 0xF>(v, -INFINITY));
    v = fmax(v, dpp_fetch<0x112, 0xF>(v, -INFINITY));
    v = fmax(v, dpp_fetch<0x114, 0xF>(v, -INFINITY));
    v = fmax(v, dpp_fetch<0x118, 0xF>(v, -INFINITY));
    v = fmax(v, dpp_fetch<0x142, 0xA>(v, -INFINITY));
    v = fmax(v, dpp_fetch<0x143, 0xC>(v, -INFINITY));
    return read_lane63(v);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_fetch<0x111, 0xF>(v, 0.0);
    v += dpp_fetch<0x112, 0xF>(v, 0.0);
    v += dpp_fetch<0x114, 0xF>(v, 0.0);
    v += dpp_fetch<0x118, 0xF>(v, 0.0);
    v += dpp_fetch<0x142, 0xA>(v, 0.0);
    v += dpp_fetch<0x143, 0xC>(v, 0.0);
    return read_lane63(v);
}

// log(1 + a) for a >= 0 (+inf -> +inf, NaN -> NaN): the Student-t proposal's log1p(u^2 / df), once per latent dim and sample
// (scipy.stats.t.logpdf, main_density_est.py:301-305).  The device library's log1p covers negative arguments, denormal
// results and every rounding corner in ~105 float64 instructions; the narrow generators' passes are bound by exactly this
// kind of per-sample vector work (fp32 MFMA and VALU exclude each other), so the scorer carries its own, ~45 instructions:
//   1 + a = w + c exactly (w the rounded sum, c its rounding error);  w = m 2^e, m in [sqrt(1/2), sqrt(2));
//   log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716: 2 s (1 + z/3 + z^2/5 + ... + z^10/21), z = s^2 (next term < 1e-18);
//   log(1 + a) = e ln 2 + log m + c / w  (c / w: the first-order term of the rounding error; the second is < 2^-106).
// The quotient s comes from a reciprocal seed refined twice and one residual correction; c / w from a seed refined once.
// <= 2 ulp (3.4e-16 relative) against long-double log1p over 2e7 arguments in [1e-304, 1e304] (the same arithmetic on the
// host, scripts/log1p_check.c; on the device: rt_log1p_pos, tests/test_gpu_parity.py).  No contraction beyond the fmas
// written here: every instantiation of the fused kernel computes the same bits.
__device__ __forceinline__ double log1p_pos(double a) {
#pragma clang fp contract(off)
    const double w = 1.0 + a;
    const double c = (a >= 1.0) ? 1.0 - (w - a) : a - (w - 1.0);
    double m = __builtin_amdgcn_frexp_mant(w);                     // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(w);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = (double)__builtin_amdgcn_rcpf((float)d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    double s = f * r;
    s = __builtin_fma(r, __builtin_fma(-d, s, f), s);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0); p = __builtin_fma(p, z, 1.0 / 17.0); p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0); p = __builtin_fma(p, z, 1.0 / 11.0); p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);  p = __builtin_fma(p, z, 1.0 / 5.0);  p = __builtin_fma(p, z, 1.0 / 3.0);
    double rw = (double)__builtin_amdgcn_rcpf((float)w);           // 0 beyond float's range: c / w is far below an ulp there
    rw = __builtin_fma(rw, __builtin_fma(-w, rw, 1.0), rw);
    const double ed = (double)e, s2 = s + s;
    const double res = __builtin_fma(ed, 6.93147180369123816490e-01, s2) +
                       (__builtin_fma(s2 * z, p, ed * 1.90821492927058770002e-10) + c * rw);
    return (a <= 1.7976931348623157e308) ? res : a;
}
// The same function behind a call: the four-wave kernels' register allocation inside the MLP is sensitive to the size of the
// scorer that follows it (inlined there, the 512-wide SHARE kernel lost 0.7 %); they score two to five dims per thread and
// pass, so the call costs them nothing measurable.  Same instructions, same bits.
__device__ __attribute__((noinline)) double log1p_pos_call(double a) { return log1p_pos(a); }

// One workgroup per CU, up to 512 registers per wave.  NST = 2: 64 samples per pass (hidden width <= 512).
// NST = 1: 32 samples per pass, used for hidden width 1024, whose 64-sample activation image (256 KB)
// would not fit in LDS.
// SHARE: the NP threads of a sample slot share the sample's Philox calls (chosen by the host when a sample takes more
// than one call, dx + df > 4); otherwise one thread draws the whole sample.  Same values either way (see draw_sample); a
// separate instantiation because the register allocation of the one-call kernel — the 2-D sets, the bench — is kept as is.
// NW = 1 / 2 (rt_mlp.hpp): the workgroup is one wave / two waves; grid = 4 / 2 per CU.
template <int HID, int NST, bool GIVEN, bool SHARE = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void is_fused_kernel(const IsArgs a) {
    constexpr int NTHREADS = 64 * NW;     // (shadows the namespace-scope default of 256)
    constexpr int MS = 32 * NST;          // samples per pass
    constexpr int NP = NTHREADS / MS;     // threads per sample in the scoring phase
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MlpArgs& m = a.g;
    const int dx = m.in_dim, dy = m.out_dim;
    const Carve cv = carve_for(HID, MS, m.k0, dx, dy, NTHREADS);
    char* x0 = smem + cv.x0;
    char* act = smem + cv.act;
    double* scr = (double*)(smem + cv.scr);
    double* yb = (double*)(smem + cv.yb);
    float* xh = (float*)(smem + cv.xh);
    float* ob = (float*)(smem + cv.ob);
    const float* part = (const float*)act;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int s = tid % MS, p = tid / MS;   // scoring role: sample slot s, dimension phase p

    // zero the padded rows of the x0 image once: rows >= dx are never written again
    for (int i = tid; i < m.k0 * MS / 4; i += NTHREADS) ((f32x4*)x0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // many-dimensional sets (the SHARE kernels: YearPredictionMSD 20/90, ...) and one- / two-wave workgroups, whose threads
    // score more dims each: the scorer reads the output layer's bias
    // once per output dim and thread; from LDS that is a broadcast read next to the partial sums, from global memory a
    // round trip per dim
    constexpr bool OB = SHARE || NW != 4;     // (the four-wave one-call kernel — the 2-D sets, the bench — keeps its code)
    if (OB) for (int d = tid; d < dy; d += NTHREADS) ob[d] = m.bp[m.n_layers - 1][d];
    // orders the zero-fill against the first pass's put_x0 writes to the same words: a workgroup whose first pass is a
    // leftover pass (p_main == 0, a pair range of only the last entry, p_main < grid) meets no other barrier before them
    __syncthreads();

#ifdef RT_STAMPS
    Stamps st;
    for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
    st.t = __builtin_amdgcn_s_memtime();
#endif
    if (a.wg_times && tid == 0) a.wg_times[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    // this workgroup's share of the main passes and of the leftover passes (a leftover pass costs a few percent more
    // than a main one — per-slot query points — so they are dealt evenly too, not left to the last workgroups).
    // Workgroups are dispatched to the eight XCDs round-robin, and the XCDs of one chip do not run at one clock (1-2 %
    // apart, stable from launch to launch): class x = blockIdx % 8 gets the share [bal[x], bal[x+1]) of the passes, which
    // balance_update_kernel sets from how long each class took in the launch before; inside a class, equal shares.
    const int64_t n_left = a.p_total - a.p_main;
    int64_t m_lo, m_hi, l_lo, l_hi;
    if (a.bal) {
        const int x = blockIdx.x & 7, per = gridDim.x >> 3;
        const int64_t i = blockIdx.x >> 3;
        const double f0 = a.bal[x], f1 = a.bal[x + 1];              // bal[0] = 0 and bal[8] = 1 exactly
        const int64_t M0 = (int64_t)((double)a.p_main * f0), M1 = (int64_t)((double)a.p_main * f1);
        const int64_t L0 = (int64_t)((double)n_left * f0), L1 = (int64_t)((double)n_left * f1);
        m_lo = M0 + (M1 - M0) * i / per; m_hi = M0 + (M1 - M0) * (i + 1) / per;
        l_lo = L0 + (L1 - L0) * i / per; l_hi = L0 + (L1 - L0) * (i + 1) / per;
    } else {
        m_lo = a.p_main * (int64_t)blockIdx.x / gridDim.x; m_hi = a.p_main * ((int64_t)blockIdx.x + 1) / gridDim.x;
        l_lo = n_left * (int64_t)blockIdx.x / gridDim.x; l_hi = n_left * ((int64_t)blockIdx.x + 1) / gridDim.x;
    }
    const int64_t n_mine = (m_hi - m_lo) + (l_hi - l_lo);
    // (query point, pass of that point — an absolute entry index in [j_lo, j_lo + ppp)) of the current main pass,
    // advanced incrementally.  (Narrowing this loop state to
    // 32 bits was tried: the register allocator then moves 21 more values between VGPRs and AGPRs inside the MLP, +0.9 %.)
    int64_t n_u = 0, j_u = 0;
    if (m_lo < m_hi) { n_u = m_lo / a.ppp; j_u = a.j_lo + (m_lo - n_u * a.ppp); }
    int64_t n_loaded = -1;                  // query point whose y / xhat sit in LDS
    auto put_x0 = [&](int d, float v) __attribute__((always_inline)) { *(float*)(x0 + act_off<MS>(d >> 3, s, (d >> 2) & 1) + 4 * (d & 3)) = v; };

    for (int64_t ps = 0; ps < n_mine; ++ps) {
        const bool left = ps >= m_hi - m_lo;            // uniform over the workgroup
        const int64_t g_u = l_lo + (ps - (m_hi - m_lo));   // leftover pass index
        // a slot's query point, sample index and whether it carries a sample at all.  Leftover passes: slot s_ carries
        // sample ppp*MS + s_ % rem of the (s_ / rem)-th query point of the pass.  Computed from uniform state and the slot
        // number alone, and computed AGAIN after the MLP (from a laundered copy of the slot number, so that the compiler
        // cannot keep the first results alive): nothing per-thread but tid stays in registers across mlp_core.
        struct Slot { int64_t n, si; bool valid; };
        auto slot_of = [&](int s_) __attribute__((always_inline)) {
            Slot r;
            if (!left) { r.n = n_u; r.si = j_u * MS + s_; r.valid = r.si < a.S; }
            else {
                const int pt = s_ / a.rem, off = s_ - pt * a.rem;
                r.n = g_u * a.q + pt; r.si = a.left_s0 + off; r.valid = pt < a.q && r.n < a.N;
            }
            return r;
        };
        if (!left && n_u != n_loaded) {
            // the previous pass's readers of yb / xh are past the barrier that ends its scoring phase
            for (int d = tid; d < dy; d += NTHREADS) yb[d] = a.y[n_u * dy + d];
            for (int d = tid; d < dx; d += NTHREADS) xh[d] = a.xhat[n_u * dx + d];
            n_loaded = n_u;
            __syncthreads();
        }
        ST(7)
        MlpPre<HID, NW> pre;
        mlp_prefetch<HID, NST, false, NW>(m, pre, wave, lane);
        // ---- A. proposal samples -> x0 image --------------------------------------------------
        {
            const Slot sl = slot_of(s);
            if (GIVEN) {
                const double* src = a.xs + (sl.valid ? ((size_t)sl.n * a.S + sl.si) * dx : 0);
                for (int d = p; d < dx; d += NP) put_x0(d, sl.valid ? (float)src[d] : 0.0f);
            } else if (SHARE) {
                if (!sl.valid) { for (int d = p; d < dx; d += NP) put_x0(d, 0.0f); }
                else if (!left) draw_sample(a.sa, (uint64_t)(a.y_index_base + sl.n), (uint32_t)sl.si, xh, dx, put_x0, p, NP);
                else draw_sample(a.sa, (uint64_t)(a.y_index_base + sl.n), (uint32_t)sl.si, a.xhat + sl.n * dx, dx, put_x0, p, NP);
            } else if (tid < MS) {
                if (!sl.valid) { for (int d = 0; d < dx; ++d) put_x0(d, 0.0f); }
                else if (!left) draw_sample(a.sa, (uint64_t)(a.y_index_base + sl.n), (uint32_t)sl.si, xh, dx, put_x0);
                else draw_sample(a.sa, (uint64_t)(a.y_index_base + sl.n), (uint32_t)sl.si, a.xhat + sl.n * dx, dx, put_x0);
            }
        }
        __syncthreads();
        ST(0)
        // ---- B. G forward ---------------------------------------------------------------------
        mlp_core<HID, NST, false, NW, NW == 1>(m, pre, x0, act, wave, lane ST_ARG);   // one wave: quarter sums added in registers
        // ---- C. score: thread (s, p) handles dims d = p, p+NP, ... -----------------------------
        int s2 = s;
        asm volatile("" : "+v"(s2));
        const Slot sl = slot_of(s2);
        const bool valid = sl.valid;
        const int64_t n_c = valid ? sl.n : 0;           // in-bounds row for the loads of slots that carry nothing
        // The sums over dims are defined on NPH = 256 / MS PHASES: phase ph adds its dims ph, ph + NPH, ... in ascending
        // order, and the phases' sums are added in phase order (phase D).  In the four-wave kernel a thread is a phase
        // (NP = NPH); a thread of the one- / two-wave kernels carries NPH / NP phases, each in its own accumulator — the
        // same additions in the same order, so the same bits whatever the workgroup shape.
        constexpr int NPH = 256 / MS, PPT = NPH / NP;
        static_assert(PPT >= 1 && PPT * NP == NPH, "phases per thread");
        auto score = [&](const auto* yrow, const auto* centre, const int ph, double& se, double& lw) __attribute__((always_inline)) {
            int d0 = ph;
            if constexpr (OB) {
                // many output dims per thread: four at a time, every LDS read of the four issued before the first use
                // (one read-latency per four dims instead of per dim); same terms added in the same order
                for (; d0 + 3 * NPH < dy; d0 += 4 * NPH) {
                    float g[4];
                    double yv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        g[i] = mlp_out_nobias<MS>(m, part, d0 + i * NPH, s) + ob[d0 + i * NPH];
                        yv[i] = (double)yrow[d0 + i * NPH];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const double diff = yv[i] - (double)g[i]; se += diff * diff; }
                }
            }
            for (int d = d0; d < dy; d += NPH) {
                const float g_d = OB ? mlp_out_nobias<MS>(m, part, d, s) + ob[d] : mlp_out<MS>(m, part, d, s);
                const double diff = (double)yrow[d] - (double)g_d;
                se += diff * diff;
            }
            for (int d = ph; d < dx; d += NPH) {
                double x;
                if (GIVEN) x = valid ? a.xs[((size_t)n_c * a.S + sl.si) * dx + d] : 0.0;
                else x = (double)*(const float*)(x0 + act_off<MS>(d >> 3, s, (d >> 2) & 1) + 4 * (d & 3));
                const double u = ((double)centre[d] - x) * a.inv_scale;
                double l1p;
                if constexpr (NW == 4) l1p = log1p_pos_call(u * u * a.inv_df); else l1p = log1p_pos(u * u * a.inv_df);
                lw += a.half_dfp1 * l1p - 0.5 * x * x;
            }
        };
        [[maybe_unused]] double se_r = 0.0, lw_r = 0.0;   // one-wave workgroups: the sample's sums over all dims, in registers
        if constexpr (PPT == NPH) {
            // One-wave workgroups: the thread owns all NPH phases of its sample, so it walks the dims ONCE in ascending
            // order, dim d into phase d % NPH's accumulator — every phase still adds its dims ph, ph + NPH, ... in
            // ascending order — eight dims at a time with every LDS read of the eight ahead of the first use, and adds
            // the phases in phase order in registers: no LDS round trip of the phase sums, no barrier.  Same bits.
            static_assert(NPH == 4, "64-sample passes");
            double se[NPH] = {0.0, 0.0, 0.0, 0.0}, lw[NPH] = {0.0, 0.0, 0.0, 0.0};
            auto score_all = [&](const auto* yrow, const auto* centre) __attribute__((always_inline)) {
                auto chunk = [&](auto ch, const int d, const int cnt) __attribute__((always_inline)) {   // d % 4 == 0
                    constexpr int CH = decltype(ch)::value;
                    float g[CH];
                    double yv[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const bool on = i < cnt;
                        g[i] = on ? mlp_out_nobias<MS, 1>(m, part, d + i, s) + ob[d + i] : 0.0f;
                        yv[i] = on ? (double)yrow[d + i] : 0.0;
                    }
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (i < cnt) { const double diff = yv[i] - (double)g[i]; se[i & 3] += diff * diff; }
                };
                int d = 0;
                for (; d + 8 <= dy; d += 8) chunk(std::integral_constant<int, 8>{}, d, 8);
                if (d + 4 <= dy) { chunk(std::integral_constant<int, 4>{}, d, 4); d += 4; }
                if (d < dy) chunk(std::integral_constant<int, 4>{}, d, dy - d);
                for (int d4 = 0; d4 < dx; d4 += 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int dd = d4 + i;
                        if (dd < dx) {
                            double x;
                            if (GIVEN) x = valid ? a.xs[((size_t)n_c * a.S + sl.si) * dx + dd] : 0.0;
                            else x = (double)*(const float*)(x0 + act_off<MS>(dd >> 3, s, (dd >> 2) & 1) + 4 * (dd & 3));
                            const double u = ((double)centre[dd] - x) * a.inv_scale;
                            const double l1p = log1p_pos(u * u * a.inv_df);
                            lw[i] += a.half_dfp1 * l1p - 0.5 * x * x;
                        }
                    }
                }
            };
            if (!left) score_all((const double*)yb, (const float*)xh);
            else score_all(a.y + n_c * dy, a.xhat + n_c * dx);
            se_r = ((se[0] + se[1]) + se[2]) + se[3];
            lw_r = ((lw[0] + lw[1]) + lw[2]) + lw[3];
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int ph = p + q * NP;
                double se = 0.0, lw = 0.0;
                if (!left) score((const double*)yb, (const float*)xh, ph, se, lw);
                else score(a.y + n_c * dy, a.xhat + n_c * dx, ph, se, lw);
                scr[ph * MS + s] = se;
                scr[NPH * MS + ph * MS + s] = lw;
            }
            __syncthreads();
        }
        // ---- D. this pass's log-sum-exp pairs (wave 0; lane = sample slot) ----------------------
        if (wave == 0) {
            double se_t = 0.0, lw_t = 0.0;
            if constexpr (PPT == NPH) { se_t = se_r; lw_t = lw_r; }
            else if (lane < MS) {
                se_t = scr[lane]; lw_t = scr[NPH * MS + lane];
#pragma unroll
                for (int qq = 1; qq < NPH; ++qq) { se_t += scr[qq * MS + lane]; lw_t += scr[NPH * MS + qq * MS + lane]; }
            }
            const bool live = lane < MS && valid;
            for (int k = 0; k < a.n_sd; ++k) {
                double v = a.c_all[k] + (lw_t - se_t * a.inv_2sd2[k]);
                // a score of -inf (G overflowed to inf: exp(-inf) = 0) carries weight 0; NaN or +inf makes the
                // reference's estimate of the point NaN (exp(v - max) at main_density_est.py:342, the product at :344),
                // and the pair says so: (NaN, NaN), which is_finalize_kernel carries through
                const unsigned long long bad = __builtin_amdgcn_ballot_w64(live && !(v < INFINITY));
                if (!(live && v < INFINITY)) v = -INFINITY;                   // empty slot (or bad, flagged above)
                if (!left) {
                    const double M = wave_max(v);
                    const double t = wave_sum((v > -INFINITY) ? exp(v - M) : 0.0);
                    if (lane == 0) {
                        double* dst = a.partial + 2 * ((n_u * a.E + j_u) * a.n_sd + k);
                        dst[0] = bad ? (double)NAN : M; dst[1] = bad ? (double)NAN : t;
                    }
                } else {
                    // lane i < q owns the i-th query point of the pass: slots i*rem .. i*rem + rem - 1
                    double M = -INFINITY;
                    for (int o = 0; o < a.rem; ++o) M = fmax(M, __shfl(v, (lane * a.rem + o) & 63));
                    double t = 0.0;
                    for (int o = 0; o < a.rem; ++o) {
                        const double w = __shfl(v, (lane * a.rem + o) & 63);
                        t += (w > -INFINITY) ? exp(w - M) : 0.0;
                    }
                    const int64_t n_o = g_u * a.q + lane;
                    if (lane < a.q && n_o < a.N) {
                        // rem <= MS/2 <= 32 and lane * rem + rem <= MS in packed mode: the shifts are in range
                        const bool mine_bad = ((bad >> (lane * a.rem)) & ((1ull << a.rem) - 1ull)) != 0;
                        double* dst = a.partial + 2 * ((n_o * a.E + (a.E - 1)) * a.n_sd + k);
                        dst[0] = mine_bad ? (double)NAN : M; dst[1] = mine_bad ? (double)NAN : t;
                    }
                }
            }
        }
        if (!left && ++j_u == a.j_lo + a.ppp) { j_u = a.j_lo; ++n_u; }
    }
    if (a.wg_times && tid == 0) a.wg_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef RT_STAMPS
#if RT_STAMPS == 2
    st.acc[0] = __builtin_amdgcn_s_memtime() - st.t;
#endif
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < NSTAMP; ++i) atomicAdd(&g_stamps[wave * NSTAMP + i], st.acc[i]);
#if RT_STAMPS == 2
    if (tid == 0 && blockIdx.x < 1024) {
        g_wg_ticks[2 * blockIdx.x] = st.acc[0];
        g_wg_ticks[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, common to all CUs: end time
    }
#endif
#endif
}

// XCD balance, closed on the device: after a (long enough) launch of the fused kernel one wave turns the workgroups'
// start / end times (s_memrealtime: 100 MHz, common to the chip) into the shares of the NEXT launch.  Class x did the
// share w_x = bal[x+1] - bal[x] of the passes in T_x = its last workgroup's end - the first workgroup's start, so it runs
// at a rate proportional to w_x / T_x, and shares proportional to those rates make the classes finish together.  Shares
// are clamped to [0.8, 1.25] / 8; the cumulative table keeps bal[0] = 0 and bal[8] = 1 exactly.  Stream order makes the
// table safe to read in the next launch's prologue.  The work a query point gets does not depend on any of this: only
// WHICH workgroup runs a pass changes, and every pass writes its own pair.
__global__ __launch_bounds__(64) void balance_update_kernel(double* bal, const unsigned long long* wg_times, int grid) {
    const int lane = threadIdx.x;
    unsigned long long t0 = ~0ull;
    for (int b = lane; b < grid; b += 64) { const unsigned long long v = wg_times[2 * b]; t0 = v < t0 ? v : t0; }
    for (int o = 32; o; o >>= 1) { const unsigned long long v = __shfl_xor(t0, o); t0 = v < t0 ? v : t0; }
    unsigned long long te = 0;                                   // lanes x, x+8, ...: the latest end of class x
    for (int b = lane; b < grid; b += 64) { const unsigned long long v = wg_times[2 * b + 1]; te = v > te ? v : te; }
    for (int o = 32; o >= 8; o >>= 1) { const unsigned long long v = __shfl_xor(te, o); te = v > te ? v : te; }
    const double T = (double)(te - t0);                          // lane x < 8: T_x (grid % 8 == 0, so class = lane % 8)
    const double w = (lane < 8) ? bal[lane + 1] - bal[lane] : 0.0;
    double r = (lane < 8 && T > 0.0) ? w / T : 0.0;
    double sum = r;
    for (int o = 4; o; o >>= 1) sum += __shfl_xor(sum, o);      // over lanes 0..7 (lanes >= 8 carry 0 and are ignored)
    double nw = (sum > 0.0) ? r / sum : 0.125;
    nw = fmin(fmax(nw, 0.8 / 8.0), 1.25 / 8.0);
    double tot = nw;
    for (int o = 4; o; o >>= 1) tot += __shfl_xor(tot, o);
    nw /= tot;
    double cum = nw;                                             // inclusive prefix over lanes 0..7
    for (int o = 1; o < 8; o <<= 1) { const double v = __shfl_up(cum, o); if ((lane & 7) >= o) cum += v; }
    if (lane < 7) bal[lane + 1] = cum;                           // bal[0] and bal[8] stay 0 and 1
}

// out[k*out_stride + n] for sd index k: merges the E pairs of query point n (n < N, this launch's block) in a fixed
// order — one wave per point, lane l takes entries l, l+64, ..., then a butterfly over the lanes.
__global__ __launch_bounds__(256) void is_finalize_kernel(const double* partial, int64_t N, int64_t E, int n_sd,
                                                          double logS, int log_flag, double* out, int64_t out_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;                     // the whole wave leaves together
    const int64_t cs = 2 * n_sd;            // entry stride in doubles
    for (int k = 0; k < n_sd; ++k) {
        const double* p = partial + 2 * (n * E * n_sd + k);
        double M = -INFINITY;
        for (int64_t e = lane; e < E; e += 64) M = fmax(M, p[e * cs]);   // fmax skips the NaN of a flagged pair
        M = wave_max(M);
        double t = 0.0;
        for (int64_t e = lane; e < E; e += 64)
            if (!(p[e * cs] == -INFINITY)) t += p[e * cs + 1] * exp(p[e * cs] - M);   // a (NaN, NaN) pair makes t NaN
        t = wave_sum(t);
        // every weight 0: the reference's log branch forms (-inf) - (-inf) = NaN (main_density_est.py:342), its linear
        // branch the mean of zeros (:345)
        double lp = (M > -INFINITY) ? M + log(t) - logS : (log_flag ? (double)NAN : -INFINITY);
        if (t != t) lp = (double)NAN;
        if (lane == 0) out[(int64_t)k * out_stride + n] = log_flag ? lp : exp(lp);
    }
}

// ----------------------------------------------------------------------------------------------
// Plain MLP forward (predict_x / predict_y): 32*NST rows per pass.
// ----------------------------------------------------------------------------------------------
template <int HID, int NST, int NW = 4>
__global__ __launch_bounds__(64 * NW) void mlp_fwd_kernel(const MlpArgs m, const float* in, int64_t N,
                                                          float* out) {
    constexpr int NTHREADS = 64 * NW;
    constexpr int MS = 32 * NST;
    constexpr int NP = NTHREADS / MS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Carve cv = carve_for(HID, MS, m.k0, m.in_dim, m.out_dim, NTHREADS);
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
        MlpPre<HID, NW> pre;
        mlp_prefetch<HID, NST, false, NW>(m, pre, wave, lane);
        mlp_core<HID, NST, false, NW>(m, pre, x0, act, wave, lane ST_ARG);
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
template <int HID, int NW = 4>
__global__ __launch_bounds__(64 * NW) void mlp_jvp_kernel(const MlpArgs m, const float* zero_bias, const float* in, int64_t N,
                                                          float* yout, float* J) {
    constexpr int NTHREADS = 64 * NW;
    constexpr int MS = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Carve cv = carve_for(HID, MS, m.k0, m.in_dim, m.out_dim, NTHREADS);
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
        for (int d = ph; d < m.in_dim; d += NW) {
            float v;
            if (sl < 32) v = (row < N) ? in[row * m.in_dim + d] : 0.0f;
            else v = (d == dir) ? 1.0f : 0.0f;
            *(float*)(x0 + act_off<MS>(d >> 3, sl, (d >> 2) & 1) + 4 * (d & 3)) = v;
        }
        __syncthreads();
#ifdef RT_STAMPS
        Stamps st; st.t = 0; for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
#endif
        MlpPre<HID, NW> pre;
        mlp_prefetch<HID, 2, 1, NW>(m, pre, wave, lane, zero_bias);
        mlp_core<HID, 2, 1, NW>(m, pre, x0, act, wave, lane ST_ARG, nullptr, zero_bias);
        if (row < N) {
            if (sl < 32) {
                if (dir == 0 && yout)
                    for (int d = ph; d < m.out_dim; d += NW) yout[row * m.out_dim + d] = mlp_out<MS>(m, part, d, sl);
            } else {
                for (int d = ph; d < m.out_dim; d += NW)
                    J[(row * m.out_dim + d) * m.in_dim + dir] = mlp_out_nobias<MS>(m, part, d, sl);
            }
        }
    }
}

// The same Jacobian with the primal evaluated ONCE per block of 32 rows instead of once per input direction, as two
// kernels.  mlp_jvp_first_kernel: one pass per block of rows — mlp_jvp_kernel's pass for direction 0 (primal + tangent e_0) —
// which also leaves, per hidden layer and thread, the 64-bit mask of its positive primal pre-activations
// (mask_ws[block of rows][hidden layer][thread]).  mlp_jvp_rest_kernel: the remaining (block of rows, direction >= 1)
// items in order, TWO per pass — item 2p in sample tile 0, item 2p+1 in tile 1, each tile's leaky-ReLU slopes taken from
// its own block's masks (for an even in_dim a pass straddles two blocks of rows) — with all-zero hidden biases (m0).
// Every pass of all three kernels runs the fused kernel's skewed layer schedule (rt_mlp.hpp skew_layer, JM: tile B two
// k-groups behind tile A, the mask select / leaky ReLU as the one VALU cluster per tile); passes are dealt one by one
// like everywhere else.  (1 + in_dim) / 2 passes per block of rows instead of in_dim: the same MFMA chains on the same
// operands, so J has the same bits.  (One kernel doing both kinds of pass spilled registers at width 512.)
template <int HID, int NW = 4>
__global__ __launch_bounds__(64 * NW) void mlp_jvp_first_kernel(const MlpArgs m, const float* zero_bias, const float* in,
                                                                int64_t N, float* yout, float* J,
                                                                unsigned long long* mask_ws) {
    constexpr int NTHREADS = 64 * NW;
    constexpr int MS = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Carve cv = carve_for(HID, MS, m.k0, m.in_dim, m.out_dim, NTHREADS);
    char* x0 = smem + cv.x0;
    char* act = smem + cv.act;
    const float* part = (const float*)act;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    for (int i = tid; i < m.k0 * MS / 4; i += NTHREADS) ((f32x4*)x0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t nblk = (N + 31) / 32;
    const int sl = tid & 63, ph = tid >> 6;       // column slot, dimension phase
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row = blk * 32 + (sl & 31);
        __syncthreads();
        for (int d = ph; d < m.in_dim; d += NW) {
            float v;
            if (sl < 32) v = (row < N) ? in[row * m.in_dim + d] : 0.0f;
            else v = (d == 0) ? 1.0f : 0.0f;
            *(float*)(x0 + act_off<MS>(d >> 3, sl, (d >> 2) & 1) + 4 * (d & 3)) = v;
        }
        __syncthreads();
#ifdef RT_STAMPS
        Stamps st; st.t = 0; for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
#endif
        MlpPre<HID, NW> pre;
        mlp_prefetch<HID, 2, 1, NW>(m, pre, wave, lane, zero_bias);
        mlp_core<HID, 2, 1, NW>(m, pre, x0, act, wave, lane ST_ARG, mask_ws + (size_t)blk * (m.n_layers - 1) * NTHREADS,
                                zero_bias);
        if (row < N) {
            if (sl < 32) {
                if (yout)
                    for (int d = ph; d < m.out_dim; d += NW) yout[row * m.out_dim + d] = mlp_out<MS>(m, part, d, sl);
            } else {
                for (int d = ph; d < m.out_dim; d += NW)
                    J[(row * m.out_dim + d) * m.in_dim] = mlp_out_nobias<MS>(m, part, d, sl);
            }
        }
    }
}

template <int HID, int NW = 4>
__global__ __launch_bounds__(64 * NW) void mlp_jvp_rest_kernel(const MlpArgs m0, int64_t N, float* J,
                                                               const unsigned long long* mask_ws) {
    constexpr int NTHREADS = 64 * NW;
    constexpr int MS = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Carve cv = carve_for(HID, MS, m0.k0, m0.in_dim, m0.out_dim, NTHREADS);
    char* x0 = smem + cv.x0;
    char* act = smem + cv.act;
    const float* part = (const float*)act;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    for (int i = tid; i < m0.k0 * MS / 4; i += NTHREADS) ((f32x4*)x0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nd = m0.in_dim - 1;                                  // directions 1 .. in_dim-1 of every block of rows
    const int64_t nitem = ((N + 31) / 32) * nd;                    // (block, direction) items, two per pass
    const int64_t npass = (nitem + 1) / 2;
    const int sl = tid & 63, ph = tid >> 6;
    const size_t mask_stride = (size_t)(m0.n_layers - 1) * NTHREADS;
    for (int64_t ps = blockIdx.x; ps < npass; ps += gridDim.x) {
        const int64_t itA = 2 * ps, blkA = itA / nd;
        const int dirA = 1 + (int)(itA - blkA * nd);
        const bool hasB = itA + 1 < nitem;                         // the very last pass of an odd item count: tile 1 idles
        const int64_t blkB = (dirA < nd || !hasB) ? blkA : blkA + 1;
        const int dirB = !hasB ? m0.in_dim : (dirA < nd ? dirA + 1 : 1);
        const int mydir = (sl < 32) ? dirA : dirB;                 // this column's direction (== in_dim: an idle tile)
        const int64_t row = ((sl < 32) ? blkA : blkB) * 32 + (sl & 31);
        __syncthreads();
        for (int d = ph; d < m0.in_dim; d += NW)
            *(float*)(x0 + act_off<MS>(d >> 3, sl, (d >> 2) & 1) + 4 * (d & 3)) = (d == mydir) ? 1.0f : 0.0f;
        __syncthreads();
#ifdef RT_STAMPS
        Stamps st; st.t = 0; for (int i = 0; i < NSTAMP; ++i) st.acc[i] = 0;
#endif
        MlpPre<HID, NW> pre;
        mlp_prefetch<HID, 2, 2, NW>(m0, pre, wave, lane);
        mlp_core<HID, 2, 2, NW>(m0, pre, x0, act, wave, lane ST_ARG,
                                const_cast<unsigned long long*>(mask_ws) + (size_t)blkA * mask_stride, nullptr,
                                mask_ws + (size_t)blkB * mask_stride);
        if (row < N && mydir < m0.in_dim)
            for (int d = ph; d < m0.out_dim; d += NW)
                J[(row * m0.out_dim + d) * m0.in_dim + mydir] = mlp_out_nobias<MS>(m0, part, d, sl);
    }
}

// ----------------------------------------------------------------------------------------------
// Laplace (closed-form) estimator, the part after the three network evaluations (main_density_est.py:365-384; v2
// roundtrip.py:341-366).  One wave per query point; everything in float64, like the reference's numpy:
//   r = y - G(xhat),  rt_error = |r|^2                                                      (:365)
//   A = J^T J,  b = J^T r                                                                   (:371, :373)
//   Sigma_inv = I + A / sd^2,  Sigma = inv(Sigma_inv)                                       (:375-376)
//   mu = Sigma (b / sd^2 - xhat)                                                            (:378)
//   c(y) = |xhat|^2 + rt_error / sd^2 - mu^T Sigma_inv mu                                   (:380)
//   log p = -dy log(sqrt(2 pi) sd) + 0.5 log det Sigma - 0.5 c                              (:382)
//   p     = 1 / (sqrt(2 pi) sd)^dy * sd^dy * sqrt(det Sigma) * exp(-0.5 c)                  (:384, extra sd^dy kept)
// Sigma_inv is symmetric positive definite (eigenvalues >= 1), so instead of the reference's LU-based inv / det the
// wave factors it once, Sigma_inv = L L^T (Cholesky, in LDS): det Sigma = prod 1 / L_kk^2, and with w = L^-1 (b / sd^2 -
// xhat) (forward substitution, fused into the factorisation's column loop) mu^T Sigma_inv mu = |w|^2 — the same numbers
// without ever forming Sigma or mu.  J arrives as the float32 values get_jacobian returns (:254: a float64 container of
// float32 results), G(xhat) and xhat likewise (:225, :240).
// lin_scale = 1 / (sqrt(2 pi) sd)^dy * sd^dy and c0 = -dy log(sqrt(2 pi) sd) come from the host.
// LDS: dx*dx + dy + dx doubles; STAGE: + the point's Jacobian, dy*dx floats, copied in with coalesced loads first — the
// dx(dx+1)/2 dot products then read it from LDS instead of issuing two scattered global loads per multiply-add, which
// is what bounded this kernel (YPMSD 20/90: 2.03 -> 0.86 ms per 37 k points).  Only the lower triangle of Sigma_inv is formed:
// the factorisation reads nothing else.
// ----------------------------------------------------------------------------------------------
template <bool STAGE>
__global__ __launch_bounds__(64) void cf_finish_kernel(const double* y, const float* xhat, const float* gy, const float* J,
                                                       int64_t N, int dx, int dy, double sd2, double c0, double lin_scale,
                                                       int log_flag, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* M = (double*)smem;          // dx x dx, lower triangle used
    double* r = M + dx * dx;            // dy
    double* w = r + dy;                 // dx
    float* Js = (float*)(w + dx);       // dy x dx (STAGE)
    const int lane = threadIdx.x;
    const int ntri = dx * (dx + 1) / 2;
    for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
        const float* Jn = J + (size_t)n * dy * dx;
        if constexpr (STAGE) {
            const int cnt = dy * dx;
            if ((cnt & 3) == 0) {
                for (int i = lane; i < cnt / 4; i += 64) ((f32x4*)Js)[i] = ((const f32x4*)Jn)[i];
            } else {
                for (int i = lane; i < cnt; i += 64) Js[i] = Jn[i];
            }
        }
        double se = 0.0, xx = 0.0;
        for (int i = lane; i < dy; i += 64) {
            const double t = y[n * dy + i] - (double)gy[n * dy + i];
            r[i] = t;
            se += t * t;
        }
        for (int d = lane; d < dx; d += 64) { const double t = (double)xhat[n * dx + d]; xx += t * t; }
        se = wave_sum(se);
        xx = wave_sum(xx);
        __syncthreads();
        const float* Jr = STAGE ? Js : Jn;
        for (int p = lane; p < ntri; p += 64) {
            int d1 = (int)((__builtin_sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);     // row of entry p of the lower triangle
            while (d1 * (d1 + 1) / 2 > p) --d1;
            while ((d1 + 1) * (d1 + 2) / 2 <= p) ++d1;
            const int d2 = p - d1 * (d1 + 1) / 2;
            double a = 0.0;
            for (int i = 0; i < dy; ++i) a = fma((double)Jr[i * dx + d1], (double)Jr[i * dx + d2], a);
            M[d1 * dx + d2] = ((d1 == d2) ? 1.0 : 0.0) + a / sd2;
        }
        for (int d = lane; d < dx; d += 64) {
            double b = 0.0;
            for (int i = 0; i < dy; ++i) b = fma((double)Jr[i * dx + d], r[i], b);
            w[d] = b / sd2 - (double)xhat[n * dx + d];
        }
        __syncthreads();
        // right-looking Cholesky of the lower triangle, column by column, with the forward substitution riding along
        double sum_log_l = 0.0, prod_l = 1.0, quad = 0.0;     // wave-uniform
        for (int k = 0; k < dx; ++k) {
            const double l = sqrt(M[k * dx + k]);
            const double wk = w[k] / l;
            sum_log_l += log(l);
            prod_l *= l;
            quad += wk * wk;
            for (int i = k + 1 + lane; i < dx; i += 64) {
                const double m = M[i * dx + k] / l;
                M[i * dx + k] = m;
                w[i] -= m * wk;
            }
            __syncthreads();
            const int m_ = dx - k - 1;
            for (int p = lane; p < m_ * m_; p += 64) {
                const int a_ = p / m_, b_ = p - a_ * m_;
                if (b_ <= a_) M[(k + 1 + a_) * dx + (k + 1 + b_)] -= M[(k + 1 + a_) * dx + k] * M[(k + 1 + b_) * dx + k];
            }
            __syncthreads();
        }
        const double c = xx + se / sd2 - quad;
        if (lane == 0) out[n] = log_flag ? c0 - sum_log_l - 0.5 * c : lin_scale * (1.0 / prod_l) * exp(-0.5 * c);
        __syncthreads();      // r / w / M are rewritten for the next point
    }
}

// ----------------------------------------------------------------------------------------------
// Mean and standard error of N float64 values — the validation / test bookkeeping of train()
// (main_density_est.py:187-188, 194-195: np.mean(py_est), np.std(py_est) / np.sqrt(len(py_est))) — on the device, in
// NUMPY'S OWN SUMMATION ORDER, so that the two numbers carry the bits numpy gives on the same array:
//   np.add.reduce over a contiguous float64 array walks it in buffers of 8192 elements (np.getbufsize()), adds each
//   buffer's pairwise sum to a running total that starts at 0, and the pairwise sum of n elements is: n < 8: a plain
//   loop from 0; n <= 128: eight strided partial sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail
//   one by one; else the sum of the first n2 = n/2 - (n/2) % 8 elements plus the sum of the rest, recursively.
//   np.mean = that sum / N; np.std = sqrt(sum((x - mean)^2) / N) with the subtraction and the product rounded
//   separately (no fused multiply-add).
// np_chunk_sum_kernel: one workgroup per buffer; thread h is node h of the buffer's recursion tree in heap order (at most
// 255 nodes: depth <= 7), leaves sum their elements, inner nodes are combined level by level.  SQDEV: the elements are
// (v - *mean_p)^2.  np_reduce_kernel (one thread) adds the buffers' sums in order and finishes the statistic.
// ----------------------------------------------------------------------------------------------
constexpr int NP_BUF = 8192;     // numpy's default ufunc buffer size, in elements
// Separately rounded float64 operations.  hipcc contracts a * b + c into an fma by default (-ffp-contract=fast-honor-pragmas),
// and HIP's __dadd_rn / __dmul_rn are plain operators in a header compiled that way, so they fuse too once inlined; the
// operators below are compiled with contraction off.
__device__ __forceinline__ double add_rn(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double sub_rn(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double div_rn(double a, double b) {
#pragma clang fp contract(off)
    return a / b;
}
template <bool SQDEV>
__global__ __launch_bounds__(256) void np_chunk_sum_kernel(const double* v, int64_t N, const double* mean_p,
                                                           double* chunk_sums) {
    __shared__ double val[256];
    const int64_t lo = (int64_t)blockIdx.x * NP_BUF;
    const int m = (int)((N - lo < NP_BUF) ? N - lo : NP_BUF);
    const int h = threadIdx.x;
    const int d = 31 - __clz(h + 1), i = h + 1 - (1 << d);      // depth and index within the level
    int off = 0, len = m;
    bool real = h < 255;
    for (int b = d - 1; b >= 0 && real; --b) {
        if (len <= 128) { real = false; break; }                  // an ancestor is a leaf: this node does not exist
        int n2 = len / 2;
        n2 -= n2 % 8;
        if ((i >> b) & 1) { off += n2; len -= n2; } else len = n2;
    }
    const double mean = SQDEV ? *mean_p : 0.0;
    auto at = [&](int k) {
        const double x = v[lo + off + k];
        if (!SQDEV) return x;
        const double t = sub_rn(x, mean);
        return mul_rn(t, t);
    };
    if (real && len <= 128) {
        double res;
        if (len < 8) {
            res = 0.0;
            for (int k = 0; k < len; ++k) res = add_rn(res, at(k));
        } else {
            double r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = at(j);
            int k = 8;
            for (; k < len - (len % 8); k += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = add_rn(r[j], at(k + j));
            }
            res = add_rn(add_rn(add_rn(r[0], r[1]), add_rn(r[2], r[3])),
                            add_rn(add_rn(r[4], r[5]), add_rn(r[6], r[7])));
            for (; k < len; ++k) res = add_rn(res, at(k));
        }
        val[h] = res;
    }
    __syncthreads();
    for (int dd = 6; dd >= 0; --dd) {
        if (real && d == dd && len > 128) val[h] = add_rn(val[2 * h + 1], val[2 * h + 2]);
        __syncthreads();
    }
    if (h == 0) chunk_sums[blockIdx.x] = val[0];
}
// Correctly rounded float64 square root (what numpy's np.sqrt returns on the host): the device's sqrt() is an approximation
// sequence that can be an ulp off.  One Newton step on the exact residual e = x - s^2 (an fma), then s moves to its
// neighbour while the root lies beyond the midpoint between them.  The midpoint test is exact: with u = sn - s (one ulp,
// signed) x - (s + u/2)^2 = (e - s u) - u^2/4, and e - s u is a multiple of u^2, so its sign (0 counted as "not beyond"
// going up, "beyond" going down) decides.  For x where s^2 would lose bits to underflow the plain result is kept.
__device__ double sqrt_cr(double x) {
#pragma clang fp contract(off)
    double s = sqrt(x);
    if (!(x > 1e-290) || !(x < 1e300)) return s;
    s = s + fma(-s, s, x) / (2.0 * s);
    for (int it = 0; it < 4; ++it) {
        const double e = fma(-s, s, x);
        if (e == 0.0) break;
        const double sn = (e > 0.0) ? nextafter(s, INFINITY) : nextafter(s, 0.0);
        const double u = sn - s;
        const double d = e - s * u;                       // s * u is exact (u is a power of two)
        const bool beyond = (e > 0.0) ? (d > 0.0) : (d <= 0.0);
        if (!beyond) break;
        s = sn;
    }
    return s;
}

// stage 0: out2[0] = mean.  stage 1 (chunk_sums of the squared deviations): out2[1] = sqrt(sum / N) / sqrt(N).
__global__ void np_reduce_kernel(const double* chunk_sums, int64_t nchunks, int64_t N, int stage, double* out2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double res = 0.0;
    for (int64_t c = 0; c < nchunks; ++c) res = add_rn(res, chunk_sums[c]);
    const double q = div_rn(res, (double)N);
    if (stage == 0) out2[0] = q;
    else out2[1] = div_rn(sqrt_cr(q), sqrt_cr((double)N));
}

// float32 feed of a float64 array: what the reference does when it hands y_points to the TF placeholder (:247)
__global__ void cast_f64_f32_kernel(const double* in, int64_t n, float* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (float)in[t];
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


// The raw generator of the sampler on caller-chosen (counter, key) words: in[6t..6t+5] = c0 c1 c2 c3 k0 k1,
// out[4t..4t+3] = philox4x32_10(...).  For the known-answer test against the published Random123 vectors.
__global__ void log1p_pos_kernel(const double* in, int64_t n, double* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (t & 1) ? log1p_pos_call(in[t]) : log1p_pos(in[t]);     // both forms the fused kernels use
}
__global__ void philox_kernel(const uint32_t* in, int64_t n, uint32_t* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const U4 r = philox4x32_10(in[6 * t], in[6 * t + 1], in[6 * t + 2], in[6 * t + 3], in[6 * t + 4], in[6 * t + 5]);
    out[4 * t] = r.x; out[4 * t + 1] = r.y; out[4 * t + 2] = r.z; out[4 * t + 3] = r.w;
}
