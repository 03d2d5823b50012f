// Microbenchmark: do the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) and the vector ALU of ONE SIMD overlap when the two
// instruction streams come from two different waves on that SIMD — and when they come from one wave in program order?
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w + 4 share SIMD w, checked through HW_ID); waves 0-3
// run role A, waves 4-7 role B; each wave times its own loop with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o build_diag/ubench_two_waves scripts/ubench_two_waves.hip && build_diag/ubench_two_waves
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { IDLE = 0, MFMA = 1, FMA32 = 2, FMA64 = 3, MIX32 = 4, MIX64 = 5, TRANS = 6 };
constexpr int PER_ITER = 8;      // MFMAs per loop iteration (two feature tiles x two sample tiles x two k-steps)
constexpr int VPER_ITER = 32;    // vector instructions per loop iteration

__device__ __forceinline__ void mfma8(f32x16 (&acc)[4], float a, float b) {
#pragma unroll
    for (int i = 0; i < PER_ITER; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
}

__global__ __launch_bounds__(512) void k(float* out, int iters, int roleA, int roleB, unsigned long long* ticks, unsigned* hwid) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? roleA : roleB;
    const float x = 1.0f + 1e-7f * threadIdx.x, y = 1e-9f * blockIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float f[8]; double d[8];
    for (int i = 0; i < 8; ++i) { f[i] = x + i; d[i] = x + i; }
    const double xd = x, yd = y;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == MFMA) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) mfma8(acc, x, y);
    } else if (role == FMA32) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < VPER_ITER; ++i) f[i & 7] = __builtin_fmaf(f[i & 7], x, y);
    } else if (role == FMA64) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < VPER_ITER; ++i) d[i & 7] = __builtin_fma(d[i & 7], xd, yd);
    } else if (role == TRANS) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < VPER_ITER; ++i) f[i & 7] = __builtin_amdgcn_exp2f(f[i & 7]);
    } else if (role == MIX32 || role == MIX64) {   // one wave, program order: 1 MFMA, then 4 independent vector instructions
#pragma unroll 1
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < PER_ITER; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (role == MIX32) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[(4 * i + j) & 7] = __builtin_fmaf(f[(4 * i + j) & 7], x, y);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[(4 * i + j) & 7] = __builtin_fma(d[(4 * i + j) & 7], xd, yd);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        ticks[wave] = t1 - t0;
        hwid[wave] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_ID bits 5:4 = SIMD
    }
}

static const char* NAME[] = {"idle", "mfma f32 32x32x2", "v_fma_f32", "v_fma_f64", "1 mfma + 4 v_fma_f32 (one wave)",
                             "1 mfma + 4 v_fma_f64 (one wave)", "v_exp_f32"};
int main() {
    float* out; unsigned long long* t; unsigned* hw; unsigned long long h[8]; unsigned hh[8];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 64); hipMalloc(&hw, 32);
    const int iters = 4000;
    const int cases[][2] = {{MFMA, IDLE}, {IDLE, FMA32}, {IDLE, FMA64}, {IDLE, TRANS}, {MFMA, MFMA}, {MFMA, FMA32}, {MFMA, FMA64},
                            {MFMA, TRANS}, {FMA32, FMA32}, {FMA64, FMA64}, {MIX32, IDLE}, {MIX64, IDLE}, {MIX32, MIX32}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(t, 0, 64);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, c[0], c[1], t, hw);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        }
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost); hipMemcpy(hh, hw, 32, hipMemcpyDeviceToHost);
        auto per = [&](int role, unsigned long long ticks) {
            const int n = (role == MFMA || role == MIX32 || role == MIX64) ? PER_ITER : VPER_ITER;
            return role == IDLE ? 0.0 : (double)ticks / ((double)iters * n);
        };
        auto unit = [](int role) { return (role == MFMA || role == MIX32 || role == MIX64) ? "mfma" : "instr"; };
        printf("A (wave 0, SIMD %u) = %-32s %7.2f cycles per %-5s | B (wave 4, SIMD %u) = %-32s %7.2f cycles per %s\n", hh[0] & 3,
               NAME[c[0]], per(c[0], h[0]), unit(c[0]), hh[4] & 3, NAME[c[1]], per(c[1], h[4]), unit(c[1]));
    }
    return 0;
}
