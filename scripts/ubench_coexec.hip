// Micro-benchmark (diagnostic, not part of the library): how many cycles does a v_mfma_f32_32x32x2_f32 slot take
// when the SAME wave issues K independent VALU / LDS instructions between consecutive MFMAs?  One wave per SIMD
// (256 threads per workgroup, one workgroup per CU), like the fused kernel.
//   hipcc --offload-arch=gfx950 -O3 -o build_diag/ubench_coexec scripts/ubench_coexec.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))

// KIND 0: v_mul_f32 (VOP3, SGPR operand) ; 1: v_max_f32 ; 2: v_pk_mul_f32 ; 3: ds_write_b128 ; 4: v_mul + v_max dependent pair
// 5: v_mov_b32 ; 6: s_nop 0 (scalar filler)
template <int K, int KIND>
__global__ __launch_bounds__(512) void bench(float* out, int iters, unsigned long long* cyc, float alpha, const float* wsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float v[16];
    f32x2 pv[8];
    f32x4 sv = {1.f, 2.f, 3.f, 4.f};
    f32x4 ld[4] = {sv, sv, sv, sv};
    const float* gp = wsrc + (size_t)(blockIdx.x * 512 + threadIdx.x) * 4;
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) pv[i] = f32x2{v[i], v[i + 8]};
    const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 255) * 16;
    const unsigned lp4 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 255) * 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            MFMA(i);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = (i * K + k) & 15;
                if (KIND == 0) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[j]) : "s"(alpha));
                if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 5) & 15]));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pv[j & 7]) : "v"(pv[(j + 3) & 7]));
                if (KIND == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lp), "v"(sv), "n"(4096 * (k & 7)));
                if (KIND == 4) {
                    float t;
                    asm volatile("v_mul_f32_e64 %0, %1, %2" : "=v"(t) : "v"(v[j]), "s"(alpha));
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(t));
                }
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 5) & 15]));
                if (KIND == 6) asm volatile("s_nop 0");
                if (KIND == 7) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ld[k & 3]) : "v"(gp), "n"(1024 * (k & 3)));
                if (KIND == 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[k & 3]) : "v"(lp), "n"(4096 * (k & 7)));
                // KIND 11: ds_max_f32 (LDS float max, no return) on the lane's own dword; 12: ds_write_b32 for comparison
                if (KIND == 11) asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(lp4), "v"(v[j]), "n"(2048 * (k & 7)));
                if (KIND == 12) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lp4), "v"(v[j]), "n"(2048 * (k & 7)));
                // KIND 9: the same K VALU instructions per MFMA on average, but in ONE cluster after every 8 MFMAs
                if (KIND == 10 && i == 7) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
            }
            if (KIND == 9 && i == 7) {
#pragma unroll
                for (int k = 0; k < 8 * K; ++k) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[k & 15]) : "s"(alpha));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    for (int i = 0; i < 4; ++i) sv += ld[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    s += sv[0] + sv[1] + sv[2] + sv[3];
    for (int i = 0; i < 8; ++i) s += pv[i][0] + pv[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);   // the slowest wave of the workgroup
}

// one wave per SIMD (256 threads) and two waves per SIMD (512 threads; a SIMD then runs twice the MFMAs)
float* g_w;
template <int K, int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)bench<K, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    double r[2];
    for (int two = 0; two < 2; ++two) {
        const int nt = two ? 512 : 256;
        bench<K, KIND><<<256, nt, 64 * 1024>>>(out, 10, cyc, 0.2f, g_w);
        hipMemset(cyc, 0, 8);
        bench<K, KIND><<<256, nt, 64 * 1024>>>(out, iters, cyc, 0.2f, g_w);
        unsigned long long c = 0;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        r[two] = (double)c / (iters * 8.0 * (two ? 2 : 1));
    }
    printf("%-28s K=%2d : %.2f ticks per MFMA (1 wave/SIMD)   %.2f (2 waves/SIMD)\n", name, K, r[0], r[1]);
}

int main(int argc, char** argv) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    hipMalloc(&g_w, (size_t)256 * 512 * 16 + 8192);
    hipMemset(g_w, 0, (size_t)256 * 512 * 16 + 8192);
    run<0, 0>("mfma only", out, cyc);
    if (argc > 1) {   // "lds": could the leaky-ReLU's max run in the LDS, behind the MFMAs? (DESIGN.md ledger)
        run<1, 12>("ds_write_b32", out, cyc);
        run<4, 12>("ds_write_b32", out, cyc);
        run<1, 11>("ds_max_f32", out, cyc);
        run<2, 11>("ds_max_f32", out, cyc);
        run<4, 11>("ds_max_f32", out, cyc);
        run<5, 11>("ds_max_f32", out, cyc);
        run<8, 11>("ds_max_f32", out, cyc);
        hipDeviceSynchronize();
        return 0;
    }
    run<1, 7>("global_load_dwordx4", out, cyc);
    run<2, 7>("global_load_dwordx4", out, cyc);
    run<1, 8>("ds_read_b128", out, cyc);
    run<2, 8>("ds_read_b128", out, cyc);
    run<1, 9>("v_mul clustered per 8 MFMA", out, cyc);
    run<4, 9>("v_mul clustered per 8 MFMA", out, cyc);
    run<8, 9>("v_mul clustered per 8 MFMA", out, cyc);
    run<1, 10>("s_waitcnt(0) per 8 MFMA", out, cyc);
    run<1, 0>("v_mul_f32_e64 (sgpr)", out, cyc);
    run<2, 0>("v_mul_f32_e64 (sgpr)", out, cyc);
    run<4, 0>("v_mul_f32_e64 (sgpr)", out, cyc);
    run<8, 0>("v_mul_f32_e64 (sgpr)", out, cyc);
    run<12, 0>("v_mul_f32_e64 (sgpr)", out, cyc);
    run<16, 0>("v_mul_f32_e64 (sgpr)", out, cyc);
    run<4, 1>("v_max_f32", out, cyc);
    run<8, 1>("v_max_f32", out, cyc);
    run<12, 1>("v_max_f32", out, cyc);
    run<2, 2>("v_pk_mul_f32", out, cyc);
    run<4, 2>("v_pk_mul_f32", out, cyc);
    run<8, 2>("v_pk_mul_f32", out, cyc);
    run<2, 4>("v_mul+v_max pair", out, cyc);
    run<4, 4>("v_mul+v_max pair", out, cyc);
    run<6, 4>("v_mul+v_max pair", out, cyc);
    run<4, 5>("v_mov_b32", out, cyc);
    run<8, 5>("v_mov_b32", out, cyc);
    run<8, 6>("s_nop", out, cyc);
    run<1, 3>("ds_write_b128", out, cyc);
    run<2, 3>("ds_write_b128", out, cyc);
    hipDeviceSynchronize();
    return 0;
}
