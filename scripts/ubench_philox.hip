// Microbenchmark: Philox4x32-10's two 32 x 32 -> 64-bit products per round as v_mul_lo_u32 + v_mul_hi_u32 (what hipcc emits
// for __umulhi and *) against one v_mad_u64_u32 each.  One wave per SIMD, a dependent chain of calls: cycles per call.
//   hipcc --offload-arch=gfx950 -O3 -o build_diag/ubench_philox scripts/ubench_philox.hip && build_diag/ubench_philox
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <bool MAD>
__device__ __forceinline__ void mul64(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    if (MAD) {
        unsigned long long p;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(b) : "vcc");
        hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
    } else { hi = __umulhi(a, b); lo = a * b; }
}
template <bool MAD>
__global__ __launch_bounds__(256) void k(uint32_t* out, int calls, unsigned long long* ticks) {
    uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = 7, c3 = 9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < calls; ++i) {
        uint32_t k0 = 1, k1 = 2;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t hi0, lo0, hi1, lo1;
            mul64<MAD>(0xD2511F53u, c0, hi0, lo0);
            mul64<MAD>(0xCD9E8D57u, c2, hi1, lo1);
            const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
int main() {
    uint32_t* out; unsigned long long* t; unsigned long long h[2];
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&t, 16);
    const int calls = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<false>, dim3(256), dim3(256), 0, 0, out, calls, t);
        hipLaunchKernelGGL(k<true>, dim3(256), dim3(256), 0, 0, out, calls, t + 1);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    uint32_t a[4], b[4];
    hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, out, 3, t); hipDeviceSynchronize(); hipMemcpy(a, out, 16, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, out, 3, t); hipDeviceSynchronize(); hipMemcpy(b, out, 16, hipMemcpyDeviceToHost);
    printf("one wave per SIMD, dependent calls: mul_lo + mul_hi %.0f cycles per Philox call, v_mad_u64_u32 %.0f; same words: %d\n",
           (double)h[0] / calls, (double)h[1] / calls, a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3]);
    return 0;
}
