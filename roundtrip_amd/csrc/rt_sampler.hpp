// rt_sampler.hpp — the counter-based proposal sampler (replaces sample_from_qx,
// /root/reference/main_density_est.py:310-318).  Included by rt_is.hip only.
#pragma once

// ----------------------------------------------------------------------------------------------
// Proposal sampler: Philox4x32-10, counter = (sample index, call index, y index lo, y index hi),
// key = seed.  Call c yields standard normals g[4c .. 4c+3] by two Box-Muller transforms.
// Integer df = ndf: chi2 = g[0]^2 + .. + g[ndf-1]^2, x_d = xhat_d + scale * g[ndf+d] / sqrt(chi2/df).
// Non-integer df: chi2 = 2*Gamma(df/2) by Marsaglia-Tsang on call indices >= 0x40000000; x_d uses g[d].
// ----------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// (k + 1/2) / 2^23 for the top 23 bits k: strictly inside (0, 1).  23 bits, not 24: k + 0.5 is exact in float32 only
// for k < 2^23 — with 24 bits the top value rounds to 2^24, u = 1, log u = 0, and one normal in 2^24 is exactly 0 (a
// chi2 of 0 for df = 1, an infinite sample, a NaN score: found in round 2 when NaN scores stopped being dropped).
__device__ __forceinline__ float u01(uint32_t x) {
    return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// Two standard normals from two uniforms (Box-Muller) on the hardware's transcendental unit: v_log_f32 is log2, so
// -2 ln u = (-2 ln 2) log2 u; v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, so the angle 2 pi u is just u;
// v_sqrt_f32.  Six quarter-rate instructions and four multiplies, where libm's logf / sincosf / sqrtf (argument
// reduction, denormal and special-case handling for inputs that cannot occur here: u is in [2^-24, 1 - 2^-24]) took
// about ninety: the sampler is per-sample work that does not shrink with the generator's width, and on a SIMD every VALU
// cycle is a lost MFMA cycle.  r >= 3.4e-4 > 0.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u01(a)));
    const float t = u01(b);
    n0 = r * __builtin_amdgcn_cosf(t);
    n1 = r * __builtin_amdgcn_sinf(t);
}

struct SamplerArgs {
    uint32_t seed_lo, seed_hi;
    int ndf;        // df if it is an integer in [1, 32], else 0 (gamma path)
    float df;
    float inv_df;   // 1 / df
    float scale;
};

__device__ float chi2_over_df_gamma(const SamplerArgs& sa, uint32_t i, uint32_t ylo, uint32_t yhi) {
    // chi2(df)/df with chi2 = 2*Gamma(df/2, 1)  (Marsaglia & Tsang 2000).
    float shape = 0.5f * sa.df;
    float boost = 1.0f;
    uint32_t call = 0x40000000u;
    if (shape < 1.0f) {
        const U4 r = philox4x32_10(i, call++, ylo, yhi, sa.seed_lo, sa.seed_hi);
        boost = powf(u01(r.x), 1.0f / shape);
        shape += 1.0f;
    }
    const float d = shape - (1.0f / 3.0f), c = 1.0f / sqrtf(9.0f * d);
    float g = d;   // fallback if 64 rejections in a row (probability < 1e-80)
    for (int it = 0; it < 64; ++it) {
        const U4 r = philox4x32_10(i, call++, ylo, yhi, sa.seed_lo, sa.seed_hi);
        float z, unused;
        box_muller(r.x, r.y, z, unused);
        float v = 1.0f + c * z;
        if (v <= 0.0f) continue;
        v = v * v * v;
        const float u = u01(r.z);
        if (logf(u) < 0.5f * z * z + d - d * v + d * logf(v)) { g = d * v; break; }
    }
    return 2.0f * g * boost / sa.df;
}

// Calls put(d, x_d) with the proposal sample (yidx, i) around xhat, for the coordinates d this thread owns.
// Philox call c yields normals 4c .. 4c+3; the first ndf of them make chi2 (every thread needs that for `mult`), the
// rest are coordinates.  The calls of one sample are independent, so `nphase` threads can share a sample: thread `phase`
// evaluates the chi2 calls plus the coordinate calls c with c % nphase == phase and puts only the latter's coordinates.
// phase = 0, nphase = 1: the whole sample.  The values do not depend on how the calls are shared.
template <typename Put>
__device__ __forceinline__ void draw_sample(const SamplerArgs& sa, uint64_t yidx, uint32_t i,
                                            const float* xhat, int dx, Put put, int phase = 0, int nphase = 1) {
    const uint32_t ylo = (uint32_t)yidx, yhi = (uint32_t)(yidx >> 32);
    // mult = scale / sqrt(chi2 / df) as scale * rsq(chi2 / df) (v_rsq_f32), the argument clamped away from 0: a chi2 of
    // exactly 0 (the hardware cosine may return 0 next to a zero crossing) would make mult infinite and 0 * inf a NaN
    // sample, which poisons the query point; clamped, such a draw lands ~1e15 scales out, G overflows and it weighs 0
    float zsum = 0.0f, mult = 0.0f;
    if (sa.ndf == 0) mult = sa.scale * __builtin_amdgcn_rsqf(fmaxf(chi2_over_df_gamma(sa, i, ylo, yhi), 1e-30f));
    const int total = sa.ndf + dx;
    for (int c = 0; 4 * c < total; ++c) {
        const bool mine = c % nphase == phase;
        if (!mine && 4 * c >= sa.ndf) continue;          // neither a chi2 call nor one of this thread's
        const U4 r = philox4x32_10(i, (uint32_t)c, ylo, yhi, sa.seed_lo, sa.seed_hi);
        float n[4];
        box_muller(r.x, r.y, n[0], n[1]);
        box_muller(r.z, r.w, n[2], n[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = 4 * c + k;
            if (g < sa.ndf) {
                zsum += n[k] * n[k];
                if (g == sa.ndf - 1) mult = sa.scale * __builtin_amdgcn_rsqf(fmaxf(zsum * sa.inv_df, 1e-30f));
            } else if (g < total && mine) {
                const int d = g - sa.ndf;
                put(d, xhat[d] + mult * n[k]);
            }
        }
    }
}
