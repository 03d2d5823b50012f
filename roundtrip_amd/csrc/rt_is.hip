// rt_is.hip — gfx950 (MI355X / CDNA4) kernels and C ABI for Roundtrip's importance-sampling density
// evaluation (reference: /root/reference/main_density_est.py:266-348, src/pyroundtrip/roundtrip.py:229-310).
//
// Design (details and numbers: DESIGN.md):
//   * one persistent 256-thread workgroup per CU; a workgroup pushes 64 proposal samples at a time
//     ("pass") through all of G's dense layers without leaving the CU;
//   * layer GEMMs run on v_mfma_f32_32x32x2_f32 in the transposed form  H_out^T = W^T · H_in^T : output
//     features are MFMA rows, samples are MFMA columns (lanes).  Wave w owns HID/4 output features x 64
//     samples (HID/128 x 2 accumulator tiles);
//   * activations live in LDS (HID x 64 fp32) in a [k/8][sample][8] image that is both the B-operand
//     read format (ds_read_b128 = four k-steps) and the accumulator write format (ds_write_b128);
//   * weights are pre-packed on the host into per-lane A-operand order, so every wave streams ITS OWN
//     feature slice straight global(L2) -> VGPR with 16-byte coalesced loads, software-pipelined one
//     k-group ahead; no LDS staging and no barrier inside a layer's K loop;
//   * the proposal sampler (Philox4x32-10 + Box-Muller), the Gaussian / Student-t scoring (fp64) and the
//     online log-sum-exp are fused around the MLP, so nothing per-sample ever reaches HBM.
//
// Nothing here falls back to the CPU: every entry point launches these kernels or returns an error.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <new>

#include "../../include/rt_is.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NTHREADS = 256;   // 4 waves, one per SIMD
constexpr int MAX_IN = 128;     // max input dim of a fused MLP (also bounded by the LDS budget)
constexpr int LDS_MAX = 160 * 1024;
constexpr int MAX_OUT = 128;    // max output dim of a fused MLP (dy for G)

struct MlpArgs {
    const float* wp[RT_MAX_LAYERS];  // packed weights, see pack_layer()
    const float* bp[RT_MAX_LAYERS];  // biases padded to a multiple of 32
    const float* wsmall;             // output layer for out_dim <= 4 (see last_layer_small), else nullptr
    int n_layers;
    int in_dim;
    int out_dim;
    int k0;       // in_dim rounded up to a multiple of 8
    int nt_out;   // ceil(out_dim / 32)
    float alpha;
};

// Diagnostic build only (-DRT_STAMPS, scripts/stamps.py): per-phase s_memtime sums of workgroup 0's waves.
// In the shipped library none of this exists.
#ifdef RT_STAMPS
constexpr int NSTAMP = 14;
__device__ unsigned long long g_stamps[4 * NSTAMP];
struct Stamps { unsigned long long t, acc[NSTAMP]; };
#define ST_PARAM , Stamps& st
#define ST_ARG , st
#if RT_STAMPS == 2   // -DRT_STAMPS=2: only the fused kernel's total (slot 0), nothing inside it is perturbed
#define ST(i)
#else
#define ST(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st.acc[i] += n_ - st.t; st.t = n_; }
#endif
#else
#define ST_PARAM
#define ST_ARG
#define ST(i)
#endif

// ----------------------------------------------------------------------------------------------
// LDS activation image.  A workgroup pass handles MS = 32*NST samples (NST = sample tiles per wave).
// Row-block kg holds k = 8kg .. 8kg+7 for the MS samples of the pass:
//   byte offset of (kg, sample s, half hh in {0,1}, element e in 0..3) =
//       ((kg*MS + s)*2 + (hh ^ swz(s)))*16 + 4e          holding k = 8kg + 4hh + e
// swz() makes the ds_read_b128 of a 32-lane half (stride 32 B between samples) and the
// ds_write_b128 of the epilogue conflict-free (bank rules: MI355X_MICROARCH.md, LDS section).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int act_swz(int s) { return ((s >> 3) ^ (s >> 2)) & 1; }
template <int MS>
__device__ __forceinline__ int act_off(int kg, int s, int hh) {
    return ((kg * MS + s) * 2 + (hh ^ act_swz(s))) * 16;
}

// ----------------------------------------------------------------------------------------------
// One k-group (8 k values = 4 MFMA k-steps of 2) for NFT feature tiles x NST sample tiles.
// MFMA 32x32x2 operand maps (cdna_hip_programming.md §3): lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// here lane half h at step e handles k = 8kg + 4h + e (any k order is a valid sum).
// ----------------------------------------------------------------------------------------------
// One v_max_f32.  fmaxf() on a value that comes straight out of an MFMA makes hipcc put a canonicalising
// v_max in front of it (128 extra instructions per epilogue); the bare instruction has fmaxf's NaN behaviour.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// lrelu of accumulator registers [4q, 4q+4) of one tile, in place: max(t, alpha*t), two elements per v_pk_mul_f32.
__device__ __forceinline__ void lrelu_quad(f32x16& t, int q, float alpha) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const f32x2 v = f32x2{t[4 * q + e], t[4 * q + e + 1]};
        const f32x2 u = v * f32x2{alpha, alpha};
        t[4 * q + e] = vmax(v[0], u[0]);
        t[4 * q + e + 1] = vmax(v[1], u[1]);
    }
}

// FIRST: the accumulators are not read — the first k-step takes the constant 0 as C, so a layer needs no
// accumulator clearing (128 v_accvgpr_write per layer otherwise).
// JVP: sample tile 1 carries tangent columns (mlp_jvp_kernel), which take no bias.
template <int NFT, int NST, bool FIRST = false, bool JVP = false>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[NFT][NST], const f32x4 (&a)[NFT],
                                           const f32x4 (&b)[NST], const f32x16* cinit = nullptr) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
            for (int st = 0; st < NST; ++st)
                acc[ft][st] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    a[ft][e], b[st][e],
                    (FIRST && e == 0) ? ((cinit && !(JVP && st == 1)) ? cinit[ft] : zero) : acc[ft][st], 0, 0, 0);
}

// Source of one wave's packed weights for a layer: a buffer resource (SGPR descriptor) plus the lane's byte offset, so
// the k-group offset is a scalar and the loads need no per-lane address arithmetic.
struct WSrc {
    __amdgpu_buffer_rsrc_t r;
    int voff;            // lane * 16
};
// base must be wave-uniform (kernel argument + wave index from readfirstlane).
__device__ __forceinline__ WSrc make_wsrc(const float* base, int lane) {
    WSrc w;
    w.r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    w.voff = lane * 16;
    return w;
}
// A operands of (k-group kg, feature tile ft); wstride = f32x4 elements between k-groups.
__device__ __forceinline__ f32x4 wload(const WSrc& w, int kg, int wstride, int ft) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // ft * 1024 goes through the SCALAR offset: as part of the vector offset hipcc hoists the four sums out of the
    // layer loop into registers of their own (AGPRs, read back with a VALU instruction before every load)
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.r, w.voff, kg * wstride * 16 + ft * 1024, 0);
    return __builtin_bit_cast(f32x4, v);
}

// Operands of k-group kg.
//   wl   : this wave's packed A operands (stride between k-groups = wstride f32x4, between feature tiles = 64 f32x4)
//   bl   : LDS byte address of this lane's B operand for (kg = 0, sample tile 0); kg stride 64*NST*16 B,
//          sample tile stride 1024 B
template <int NFT, int NST>
__device__ __forceinline__ void load_group(f32x4 (&a)[NFT], f32x4 (&b)[NST], const WSrc& wl,
                                           int wstride, const char* bl, int kg) {
    const char* l = bl + kg * (1024 * NST);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) a[ft] = wload(wl, kg, wstride, ft);
#pragma unroll
    for (int st = 0; st < NST; ++st) b[st] = *(const f32x4*)(l + 1024 * st);
}

// Short / odd K (layer 0: K = padded input dim): no software pipelining.
template <int NFT, int NST, bool JVP = false>
__device__ __forceinline__ void gemm_tile_simple(f32x16 (&acc)[NFT][NST], const WSrc& wl,
                                                 int ftt, int nkg, const char* bl, const f32x16* cinit) {
    f32x4 a[NFT], b[NST];
    load_group<NFT, NST>(a, b, wl, ftt * 64, bl, 0);
    mfma_group<NFT, NST, true, JVP>(acc, a, b, cinit);
#pragma unroll 1
    for (int kg = 1; kg < nkg; ++kg) {
        load_group<NFT, NST>(a, b, wl, ftt * 64, bl, kg);
        mfma_group<NFT, NST>(acc, a, b);
    }
}

// One pipeline stage: request the operands of a later k-group into (an, bn) and run the 4*NFT*NST MFMAs
// of the current one on (ac, bc).  hipcc would otherwise sink every load down to its first use (exposing
// the LDS / L2 latency once per group): the sched_group_barrier sequence pins "one memory instruction
// behind each of the first MFMAs", and sched_barrier(0) keeps stages from mixing.
template <int NFT, int NST, bool FIRST = false, bool JVP = false>
__device__ __forceinline__ void pipe_stage(f32x16 (&acc)[NFT][NST], f32x4 (&an)[NFT], f32x4 (&bn)[NST],
                                           const f32x4 (&ac)[NFT], const f32x4 (&bc)[NST],
                                           const WSrc& wl, int wstride, const char* bl, int kgn,
                                           const f32x16* cinit = nullptr) {
    load_group<NFT, NST>(an, bn, wl, wstride, bl, kgn);
    mfma_group<NFT, NST, FIRST, JVP>(acc, ac, bc, cinit);
    constexpr int NM = 4 * NFT * NST;   // MFMAs in the stage
#pragma unroll
    for (int i = 0; i < NFT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NFT - NST, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// Requests the weight operands of the first three k-groups of a layer.  Weights do not depend on the
// activations, so this is issued BEFORE the previous layer's epilogue and barriers and lands behind them.
template <int NFT>
__device__ __forceinline__ void prefetch_a(f32x4 (&a0)[NFT], f32x4 (&a1)[NFT], f32x4 (&a2)[NFT],
                                           const WSrc& wl, int wstride) {
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        a0[ft] = wload(wl, 0, wstride, ft);
        a1[ft] = wload(wl, 1, wstride, ft);
        a2[ft] = wload(wl, 2, wstride, ft);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// nkg must be a positive multiple of 4.  Operands of k-group kg+3 are requested while k-group kg's
// MFMAs issue.  Tail prefetches are clamped to the last group (redundant, in bounds).
// a0..a2 arrive pre-requested (prefetch_a); only the LDS operands of groups 0..2 are read here.
// Last k-group of a hidden layer (ACT0: 64-sample passes, not the Jacobian kernel).  Sample tile 0's MFMAs go first;
// its leaky-ReLU (6 VALU instructions per MFMA) then issues in the shadow of sample tile 1's MFMAs, so half of the
// epilogue's arithmetic costs no time.  Every accumulator still sees its k's in the same order.  The sched_barriers
// keep hipcc from moving the VALU work behind the MFMAs again.
template <int NFT, int NST>
__device__ __forceinline__ void final_stage_act0(f32x16 (&acc)[NFT][NST], const f32x4 (&ac)[NFT],
                                                 const f32x4 (&bc)[NST], float alpha) {
    static_assert(NST == 2, "needs two sample tiles");
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft)
            acc[ft][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ft][e], bc[0][e], acc[ft][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            acc[ft][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[ft][e], bc[1][e], acc[ft][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int idx = e * NFT + ft;            // 4*NFT chunks of one register quad each
            lrelu_quad(acc[idx >> 2][0], idx & 3, alpha);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ACT0: on return sample tile 0's accumulators already hold lrelu(pre-activation) (see final_stage_act0).
template <int NFT, int NST, bool JVP = false, bool ACT0 = false>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[NFT][NST], f32x4 (&a0)[NFT], f32x4 (&a1)[NFT],
                                          f32x4 (&a2)[NFT], const WSrc& wl,
                                          int ftt, int nkg, const char* bl, const f32x16* cinit, float alpha = 0.f) {
    f32x4 a3[NFT], b0[NST], b1[NST], b2[NST], b3[NST];
    const int wstride = ftt * 64;
    const int lastg = nkg - 1;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        b0[st] = *(const f32x4*)(bl + 1024 * st);
        b1[st] = *(const f32x4*)(bl + 1024 * NST + 1024 * st);
        b2[st] = *(const f32x4*)(bl + 2048 * NST + 1024 * st);
    }
    __builtin_amdgcn_sched_barrier(0);
    // first four k-groups peeled: their first MFMAs start the accumulators from the constant 0
    pipe_stage<NFT, NST, true, JVP>(acc, a3, b3, a0, b0, wl, wstride, bl, 3, cinit);
    pipe_stage<NFT, NST>(acc, a0, b0, a1, b1, wl, wstride, bl, min(4, lastg));
    pipe_stage<NFT, NST>(acc, a1, b1, a2, b2, wl, wstride, bl, min(5, lastg));
    pipe_stage<NFT, NST>(acc, a2, b2, a3, b3, wl, wstride, bl, min(6, lastg));
    if constexpr (!ACT0) {
#pragma unroll 1
        for (int kg = 4; kg < nkg; kg += 4) {
            pipe_stage<NFT, NST>(acc, a3, b3, a0, b0, wl, wstride, bl, kg + 3);
            pipe_stage<NFT, NST>(acc, a0, b0, a1, b1, wl, wstride, bl, min(kg + 4, lastg));
            pipe_stage<NFT, NST>(acc, a1, b1, a2, b2, wl, wstride, bl, min(kg + 5, lastg));
            pipe_stage<NFT, NST>(acc, a2, b2, a3, b3, wl, wstride, bl, min(kg + 6, lastg));
        }
    } else {
        // nkg >= 8 here (hidden layers: nkg = HID/8 >= 16); the last four k-groups are peeled
#pragma unroll 1
        for (int kg = 4; kg < nkg - 4; kg += 4) {
            pipe_stage<NFT, NST>(acc, a3, b3, a0, b0, wl, wstride, bl, kg + 3);
            pipe_stage<NFT, NST>(acc, a0, b0, a1, b1, wl, wstride, bl, kg + 4);
            pipe_stage<NFT, NST>(acc, a1, b1, a2, b2, wl, wstride, bl, kg + 5);
            pipe_stage<NFT, NST>(acc, a2, b2, a3, b3, wl, wstride, bl, kg + 6);
        }
        pipe_stage<NFT, NST>(acc, a3, b3, a0, b0, wl, wstride, bl, lastg);
        mfma_group<NFT, NST>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<NFT, NST>(acc, a2, b2);
        __builtin_amdgcn_sched_barrier(0);
        final_stage_act0<NFT, NST>(acc, a3, b3, alpha);
    }
}

// This lane's biases for a hidden layer's epilogue: features 32FT + 8q + 4h + (0..3), FT = wave*NFT + ft.
// Requested BEFORE the layer's K loop (sched_barrier keeps hipcc from sinking the loads to their use, where
// each would expose an L2 round trip), consumed by store_act after it.
template <int NFT>
__device__ __forceinline__ void load_bias(f32x16 (&bv)[NFT], const float* __restrict__ bias, int wave, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *(const f32x4*)(bias + 32 * (wave * NFT + ft) + 8 * q + 4 * h);
            bv[ft][4 * q] = t[0]; bv[ft][4 * q + 1] = t[1]; bv[ft][4 * q + 2] = t[2]; bv[ft][4 * q + 3] = t[3];
        }
    __builtin_amdgcn_sched_barrier(0);
}

// Hidden-layer epilogue: act = lrelu(acc), written in the next layer's B-operand image.  The bias is already in
// acc: the layer's first MFMA takes it as its C operand (mfma_group's cinit), so a feature's value is the fp32
// fmaf chain  fma(w_kn, x_kn, ... fma(w_k0, x_k0, bias)).
// C/D map of the 32x32 MFMA: lane (h, j): column j = sample; register r = row (r&3) + 8(r>>2) + 4h = feature.
// Registers 4q..4q+3 of tile (FT, st) are features 32FT + 8q + 4h + (0..3) = next-layer k-group 4FT+q, half h.
// JVP (NST == 2): tile 0 holds primal pre-activations t, tile 1 the tangent pre-activations dt of the SAME rows in
// the same lanes and registers, so the tangent's activation is lrelu'(t) * dt with no cross-lane traffic:
// slope 1 where t > 0, alpha elsewhere.
// ACT0: sample tile 0 is already activated (gemm_tile<..., ACT0>); only tile 1's arithmetic is left.
template <int NFT, int NST, bool JVP = false, bool ACT0 = false>
__device__ __forceinline__ void store_act(f32x16 (&acc)[NFT][NST], float alpha, char* act, int wave, int lane) {
    constexpr int MS = 32 * NST;
    const int h = lane >> 5, j = lane & 31;
    if constexpr (JVP) {
        static_assert(NST == 2, "JVP pairs sample tile 0 (primal) with sample tile 1 (tangent)");
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            const int FT = wave * NFT + ft;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[ft][0][4 * q + e], dt = acc[ft][1][4 * q + e];
                    v0[e] = vmax(t, t * alpha);
                    v1[e] = (t > 0.0f) ? dt : dt * alpha;
                }
                *(f32x4*)(act + act_off<MS>(4 * FT + q, j, h)) = v0;
                *(f32x4*)(act + act_off<MS>(4 * FT + q, 32 + j, h)) = v1;
            }
        }
        return;
    }
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        const int FT = wave * NFT + ft;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (!(ACT0 && st == 0)) lrelu_quad(acc[ft][st], q, alpha);
                const f32x4 v = {acc[ft][st][4 * q], acc[ft][st][4 * q + 1], acc[ft][st][4 * q + 2],
                                 acc[ft][st][4 * q + 3]};
                *(f32x4*)(act + act_off<MS>(4 * FT + q, 32 * st + j, h)) = v;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Skewed hidden layers (64-sample passes; not the Jacobian kernel).
//
// What the hardware allows (scripts/ubench_coexec.hip, measured on MI355X): a SIMD's fp32 MFMA and its VALU never run
// at the same time.  A VALU instruction between two MFMAs costs its full 4-5 cycles plus ~11 cycles for the
// MFMA -> VALU -> MFMA switch, whether it comes from the same wave or from a second wave on the SIMD; ds_read /
// ds_write (up to one ds_write_b128 per MFMA and wave, the LDS rate), buffer loads, scalar instructions and barriers
// between MFMAs are free.  So a layer's epilogue splits into arithmetic that must be paid for — the leaky-ReLU, as ONE
// uninterrupted VALU cluster per sample tile — and traffic that can hide behind MFMAs: the ds_write_b128 stores of the
// activation image, the next operand reads and the barriers (the image is updated in place; it cannot be
// double-buffered in 160 KB).
//
// Sample tile 1 ("B") runs SKEW = 2 k-groups behind sample tile 0 ("A") inside the same K loop: a stage s issues A's
// MFMAs for k-group s and B's for k-group s-2, with the weights of a k-group kept in registers two stages longer
// (ring of 6).  A layer is NKG + 2 stages:
//   stage 0          A only (C = bias); the PREVIOUS layer's tile B, already activated, is stored, one ds_write_b128
//                    behind each MFMA;  barrier P: B's image is published
//   stage 1          A only; B's operands of k-groups 0..2 are requested
//   stages 2..NKG-1  A and B
//                    barrier Q: every wave has read all of A's image;  A's leaky-ReLU cluster
//   stage NKG        B only; tile A is stored behind the MFMAs;  barrier R: A's image is published
//   stage NKG+1      B only; the NEXT layer's A operands of k-groups 0..2 are requested
//                    barrier S: every wave has read all of B's image;  B's leaky-ReLU cluster
// Every barrier has MFMAs in flight or a VALU cluster next to it, and no operand read waits on one.
// Every accumulator sees its k's in the same order as in the plain schedule, so results are bit-identical.
// Ring slots: weights of k-group g in w[g % 6]; A's operand of k-group g in bA[g % 6]; B's in bB[(g + 2) % 6];
// all three are requested three stages before their first use.  The stages are unrolled at compile time (skew_steady),
// so a slot is a fixed register and any NKG >= 8 works: widths 128, 256 and 512.
// ----------------------------------------------------------------------------------------------
// LDS operand reads of the skewed schedule: byte offset `off` from this lane's base bl.  ds_read_b128 carries a 16-bit
// immediate offset; with a second base at +64 KB every (compile-time) offset is an immediate and the fully unrolled
// K loop has no address arithmetic on the VALU, where a lone instruction between MFMAs costs ~15 cycles.
typedef const __attribute__((address_space(3))) char* lds_cptr;
struct LdsBase {
    lds_cptr lo;   // bl
    lds_cptr hi;   // bl + 65536
};
__device__ __forceinline__ LdsBase make_lds_base(const char* p) {
    LdsBase b;
    b.lo = (lds_cptr)p;
    unsigned hi = (unsigned)(size_t)b.lo + 65536u;
    asm volatile("" : "+v"(hi));   // opaque: or hipcc folds hi back into lo + 64 KB and adds once per read
    b.hi = (lds_cptr)(size_t)hi;
    return b;
}
__device__ __forceinline__ f32x4 lds_operand(const LdsBase& b, int off) {
    typedef const __attribute__((address_space(3))) f32x4* vp;
    return (off < 65536) ? *(vp)(b.lo + off) : *(vp)(b.hi + (off - 65536));
}

template <int NFT>
__device__ __forceinline__ void store_quad(const f32x16 (&t)[NFT][2], int st, int idx, char* st_base) {
    const int ft = idx >> 2, q = idx & 3;
    const f32x4 v = {t[ft][st][4 * q], t[ft][st][4 * q + 1], t[ft][st][4 * q + 2], t[ft][st][4 * q + 3]};
    *(f32x4*)(st_base + (4 * ft + q) * 2048 + st * 1024) = v;
}

// leaky-ReLU of a whole sample tile in place, as one VALU cluster
template <int NFT, int TILE>
__device__ __forceinline__ void lrelu_tile(f32x16 (&acc)[NFT][2], float alpha) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) lrelu_quad(acc[ft][TILE], q, alpha);
    __builtin_amdgcn_sched_barrier(0);
}

// Bias of one register quad (idx = 4*ft + q) of tile TILE, loaded straight INTO the accumulators: a layer's first MFMA
// then accumulates onto it like any other (no separate bias registers, no special first k-step).  bias: the layer's
// padded bias vector as a buffer resource; voff: this lane's byte offset (32*wave*NFT + 4h)*4.
// A bias load overwrites registers a ds_write_b128 has just read: it follows that store BLAG MFMAs later, when the store
// has long left the LDS queue (directly behind it, hipcc's WAR wait for the store stalls the wave for the LDS latency).
#ifndef RT_BLAG
#define RT_BLAG 3
#endif
template <int NFT>
constexpr int blag() { return RT_BLAG < 4 * NFT ? RT_BLAG : 4 * NFT; }
struct BiasSrc {
    __amdgpu_buffer_rsrc_t r;
    int voff;
};
__device__ __forceinline__ BiasSrc make_bias(const float* bias, int voff) {
    BiasSrc b;
    b.r = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, 0x7fffffff, 0x00020000);
    b.voff = voff;
    return b;
}
template <int NFT, int TILE>
__device__ __forceinline__ void bias_quad(f32x16 (&acc)[NFT][2], const BiasSrc& bs, int idx) {
    const int ft = idx >> 2, q = idx & 3;
    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bs.r, bs.voff, (32 * ft + 8 * q) * 4, 0));
    acc[ft][TILE][4 * q] = t[0]; acc[ft][TILE][4 * q + 1] = t[1];
    acc[ft][TILE][4 * q + 2] = t[2]; acc[ft][TILE][4 * q + 3] = t[3];
}

// The MFMAs of ring position R for both tiles.
template <int NFT, int R>
__device__ __forceinline__ void skew_mfmas(f32x16 (&acc)[NFT][2], const f32x4 (&wA)[NFT], const f32x4 (&wB)[NFT],
                                           const f32x4 (&bA)[6], const f32x4 (&bB)[6]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            acc[ft][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[ft][e], bA[R][e], acc[ft][0], 0, 0, 0);
            acc[ft][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wB[ft][e], bB[R][e], acc[ft][1], 0, 0, 0);
        }
}

// A steady stage s (ring position R = s % 6): requests the operands of three stages ahead, runs both tiles' MFMAs,
// and pins one memory instruction behind each of the first MFMAs (see pipe_stage).
//   wA / wB: the weights A and B use in this stage (ring slots, or the layer's first three k-groups in wh)
//   wld:     where the weight request goes (ring slot (R+3)%6, or the NEXT layer's wh) and from which source / k-group
//   LDA/LDB: whether tile A / tile B still has an operand to request (not in the layer's last stages)
template <int NFT, int R, bool LDA, bool LDB>
__device__ __forceinline__ void skew_stage(f32x16 (&acc)[NFT][2], const f32x4 (&wA)[NFT], const f32x4 (&wB)[NFT],
                                           f32x4 (&wld)[NFT], f32x4 (&bA)[6], f32x4 (&bB)[6],
                                           const WSrc& wl, int wstride, int kgw, const LdsBase& bl, int s) {
    constexpr int RL = (R + 3) % 6;
#ifdef RT_ABL   // timing ablation (diagnostic builds only; results are garbage): 1 = no weight loads, 2 = no LDS reads
    constexpr bool LDW_ = !(RT_ABL & 1), LDA_ = LDA && !(RT_ABL & 2), LDB_ = LDB && !(RT_ABL & 2);
#else
    constexpr bool LDW_ = true, LDA_ = LDA, LDB_ = LDB;
#endif
    if (LDW_) {
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) wld[ft] = wload(wl, kgw, wstride, ft);
    }
    if (LDA_) bA[RL] = lds_operand(bl, (s + 3) * 2048);
    if (LDB_) bB[RL] = lds_operand(bl, 1024 + (s + 1) * 2048);
    skew_mfmas<NFT, R>(acc, wA, wB, bA, bB);
    constexpr int NM = 8 * NFT;
    constexpr int NV = LDW_ ? NFT : 0, ND = (LDA_ ? 1 : 0) + (LDB_ ? 1 : 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NV - ND, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// The weights of k-group G of the current layer: its first three live in wh, the others in ring slot G % 6.
template <int G, int NFT>
__device__ __forceinline__ f32x4 (&wslot(f32x4 (&wh)[3][NFT], f32x4 (&w)[6][NFT]))[NFT] {
    if constexpr (G < 3) return wh[G];
    else return w[G % 6];
}

// Steady stages S .. NKG-1, unrolled at compile time: every LDS offset is an immediate, every weight offset a scalar,
// every ring slot a fixed register.  Stage S requests the operands of stage S+3; once those belong to the next layer
// (S+3 >= NKG) only that layer's first weights are requested, into wh.
template <int NFT, int NKG, int S>
__device__ __forceinline__ void skew_steady(f32x16 (&acc)[NFT][2], f32x4 (&wh)[3][NFT], f32x4 (&w)[6][NFT],
                                            f32x4 (&bA)[6], f32x4 (&bB)[6], const WSrc& wl, const WSrc& wn,
                                            int wstride, const LdsBase& bl) {
    if constexpr (S < NKG) {
        constexpr int KA = S + 3;
        constexpr bool NEXT = KA >= NKG;
        if constexpr (NEXT)
            skew_stage<NFT, S % 6, false, (S + 1 < NKG)>(acc, wslot<S, NFT>(wh, w), wslot<S - 2, NFT>(wh, w),
                                                         wh[KA - NKG], bA, bB, wn, wstride, KA - NKG, bl, S);
        else
            skew_stage<NFT, S % 6, true, true>(acc, wslot<S, NFT>(wh, w), wslot<S - 2, NFT>(wh, w), w[KA % 6], bA, bB,
                                               wl, wstride, KA, bl, S);
        skew_steady<NFT, NKG, S + 1>(acc, wh, w, bA, bB, wl, wn, wstride, bl);
    }
}

// A one-tile stage: the 4*NFT MFMAs of tile TILE on one k-group (weights wt, operand b); behind(idx) issues the memory
// instructions that go behind MFMA idx (stores of the other tile, operand and bias requests), pinned by sched_barriers.
template <int NFT, int TILE, class Behind>
__device__ __forceinline__ void skew_edge_stage(f32x16 (&acc)[NFT][2], const f32x4 (&wt)[NFT], const f32x4& b,
                                                Behind&& behind) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            acc[ft][TILE] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[ft][e], b[e], acc[ft][TILE], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            behind(e * NFT + ft);
            __builtin_amdgcn_sched_barrier(0);
        }
}

// One hidden layer with K = 8*NKG.  The weights of its k-groups 0..2 live in wh (outside the ring), so that the NEXT
// layer's can be requested as early as stage NKG-3 — two thousand cycles before they are used.  On entry: wh holds
// (requests for) this layer's k-groups 0..2, bA[0..2] (requests for) tile A's operands of k-groups 0..2, acc[.][0]
// (requests for) this layer's bias; tile A's input image is published; acc[.][1] holds the PREVIOUS layer's tile-B
// activations, not yet stored, and every wave is done reading tile B's image.  On return the same holds one layer on
// (wn / bias_n = the next layer's weights and bias; when there is none the caller passes this layer's again and
// ignores the result).
template <int NFT, int NKG>
__device__ __forceinline__ void skew_layer(f32x16 (&acc)[NFT][2], f32x4 (&wh)[3][NFT], f32x4 (&bA)[6],
                                           const WSrc& wl, const WSrc& wn, const BiasSrc& bias, const BiasSrc& bias_n,
                                           int wstride, const LdsBase& bl, char* st_base, float alpha ST_PARAM) {
    static_assert(NKG >= 8, "the next layer's first weights are requested into wh from stage NKG-3 on");
    f32x4 w[6][NFT], bB[6];
    // ---- stage 0: A only; previous layer's tile B stored in the shadow; requests w[3], bA[3] ---------------------
    // (this layer's bias goes into tile B's accumulators BLAG quads behind the stores that empty them)
    skew_edge_stage<NFT, 0>(acc, wh[0], bA[0], [&](int idx) {
        if (idx < NFT) w[3][idx] = wload(wl, 3, wstride, idx);
        if (idx == NFT) bA[3] = lds_operand(bl, 3 * 2048);
        store_quad<NFT>(acc, 1, idx, st_base);
        if (idx >= blag<NFT>()) bias_quad<NFT, 1>(acc, bias, idx - blag<NFT>());
    });
    ST(8)
    __syncthreads();   // P: tile B's image is published
    ST(3)
    // ---- stage 1: A only; requests w[4], bA[4], bB[2..4] -------------------------------------------------------
    skew_edge_stage<NFT, 0>(acc, wh[1], bA[1], [&](int idx) {
        if (idx < NFT) w[4][idx] = wload(wl, 4, wstride, idx);
        if (idx == 0) bA[4] = lds_operand(bl, 4 * 2048);
        if (idx == 1) bB[2] = lds_operand(bl, 1024);
        if (idx == 2) bB[3] = lds_operand(bl, 1024 + 2048);
        if (idx == 3) bB[4] = lds_operand(bl, 1024 + 4096);
        if (idx < blag<NFT>()) bias_quad<NFT, 1>(acc, bias, 4 * NFT - blag<NFT>() + idx);
    });
    ST(9)
    // ---- stages 2 .. NKG-1: A and B ------------------------------------------------------------------------------
    skew_steady<NFT, NKG, 2>(acc, wh, w, bA, bB, wl, wn, wstride, bl);
    ST(5)
    __syncthreads();   // Q: every wave has read all of tile A's image (its last operand fed stage NKG-1)
    ST(4)
    lrelu_tile<NFT, 0>(acc, alpha);
    ST(12)
    // ---- stage NKG: B only; tile A stored in the shadow ------------------------------------------------------
    //      (the next layer's bias goes into tile A's accumulators BLAG quads behind the stores that empty them)
    skew_edge_stage<NFT, 1>(acc, w[(NKG - 2) % 6], bB[NKG % 6], [&](int idx) {
        store_quad<NFT>(acc, 0, idx, st_base);
        if (idx >= blag<NFT>()) bias_quad<NFT, 0>(acc, bias_n, idx - blag<NFT>());
    });
    ST(10)
    __syncthreads();   // R: tile A's image is published
    ST(4)
    // ---- stage NKG+1: B only; requests the next layer's tile-A operands of k-groups 0..2 ---------------------------
    skew_edge_stage<NFT, 1>(acc, w[(NKG - 1) % 6], bB[(NKG + 1) % 6], [&](int idx) {
        if (idx < 3) bA[idx] = lds_operand(bl, idx * 2048);
        if (idx < blag<NFT>()) bias_quad<NFT, 0>(acc, bias_n, 4 * NFT - blag<NFT>() + idx);
    });
    ST(11)
    __syncthreads();   // S: every wave has read all of tile B's image
    ST(4)
    lrelu_tile<NFT, 1>(acc, alpha);
    ST(13)
}

// Operands of a pass that do not depend on its samples — layer 0's bias (straight into both tiles' accumulators) and
// its first weights — requested BEFORE the pass's samples are drawn, so that their round trip (they have been evicted
// from L2 by the 9 MB of weights streamed since the previous pass) hides behind the sampler.
template <int HID>
struct MlpPre {
    f32x16 acc[HID / 128][2];
    f32x4 a[HID / 128];
};
template <int HID>
__device__ __forceinline__ void skew_prefetch(const MlpArgs& m, MlpPre<HID>& pre, int wave, int lane) {
    constexpr int NFT = HID / 128, FTT = HID / 32;
    const BiasSrc b0 = make_bias(m.bp[0], (32 * wave * NFT + 4 * (lane >> 5)) * 4);
    const WSrc w0 = make_wsrc(m.wp[0] + (wave * NFT) * 256, lane);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) pre.a[ft] = wload(w0, 0, FTT * 64, ft);
#pragma unroll
    for (int idx = 0; idx < 4 * NFT; ++idx) { bias_quad<NFT, 0>(pre.acc, b0, idx); bias_quad<NFT, 1>(pre.acc, b0, idx); }
    __builtin_amdgcn_sched_barrier(0);
}

// Layer 0 and all hidden layers after it.  On return the last hidden layer's activations are in the image, the
// closing barrier still to come (the caller requests the output layer's weights first).
// Layer 0 (K = k0, a few k-groups) is the same dance without the skew: all k-groups but the last for both tiles, then
// the last one: tile A's MFMAs, A's leaky-ReLU cluster, tile B's MFMAs with A's stores behind them, barrier, B's cluster.
template <int HID>
__device__ __forceinline__ void hidden_skewed(const MlpArgs& m, MlpPre<HID>& pre, const char* x0, char* act, int wave,
                                              int lane ST_PARAM) {
    constexpr int NFT = HID / 128, FTT = HID / 32, NKG = HID / 8;
    const int h = lane >> 5, j = lane & 31;
    const int nhid = m.n_layers - 1;
    const int nkg0 = m.k0 >> 3;
    char* st_base = act + act_off<64>(4 * wave * NFT, j, h);
    const LdsBase bl = make_lds_base(act + act_off<64>(0, j, h));
    const char* xlp = x0 + act_off<64>(0, j, h);
    const int bias_voff = (32 * wave * NFT + 4 * h) * 4;
    f32x16 (&acc)[NFT][2] = pre.acc;
    f32x4 wh[3][NFT], bA[6];
    const WSrc w0 = make_wsrc(m.wp[0] + (wave * NFT) * 256, lane);
    WSrc wl = make_wsrc(m.wp[nhid > 1 ? 1 : 0] + (wave * NFT) * 256, lane);
    {   // ---- layer 0 (bias and the first weights: skew_prefetch) ----
        f32x4 (&a)[NFT] = pre.a;
        f32x4 bx[2];
        const BiasSrc b1 = make_bias(m.bp[nhid > 1 ? 1 : 0], bias_voff);
#pragma unroll 1
        for (int kg = 0; kg + 1 < nkg0; ++kg) {
            if (kg > 0) {
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) a[ft] = wload(w0, kg, FTT * 64, ft);
            }
            bx[0] = *(const f32x4*)(xlp + kg * 2048);
            bx[1] = *(const f32x4*)(xlp + 1024 + kg * 2048);
            skew_edge_stage<NFT, 0>(acc, a, bx[0], [](int) {});
            skew_edge_stage<NFT, 1>(acc, a, bx[1], [](int) {});
        }
        if (nkg0 > 1) {
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) a[ft] = wload(w0, nkg0 - 1, FTT * 64, ft);
        }
        bx[0] = *(const f32x4*)(xlp + (nkg0 - 1) * 2048);
        bx[1] = *(const f32x4*)(xlp + 1024 + (nkg0 - 1) * 2048);
        skew_edge_stage<NFT, 0>(acc, a, bx[0], [](int) {});
        // layer 1's first weights are requested only now, behind everything layer 0 waits for; they fly during the cluster
        if (nhid > 1) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) wh[g][ft] = wload(wl, g, FTT * 64, ft);
        }
        lrelu_tile<NFT, 0>(acc, m.alpha);
        // (layer 1's bias goes into tile A's accumulators BLAG quads behind the stores that empty them; harmless if
        // there is no layer 1)
        skew_edge_stage<NFT, 1>(acc, a, bx[1], [&](int idx) {
            store_quad<NFT>(acc, 0, idx, st_base);
            if (idx >= blag<NFT>()) bias_quad<NFT, 0>(acc, b1, idx - blag<NFT>());
        });
#pragma unroll
        for (int i = 0; i < blag<NFT>(); ++i) bias_quad<NFT, 0>(acc, b1, 4 * NFT - blag<NFT>() + i);
    }
    __syncthreads();   // tile A's image is published
    if (nhid > 1) {
        bA[0] = lds_operand(bl, 0);
        bA[1] = lds_operand(bl, 2048);
        bA[2] = lds_operand(bl, 4096);
    }
    lrelu_tile<NFT, 1>(acc, m.alpha);
    ST(1)
    for (int l = 1; l < nhid; ++l) {
        const int ln = (l + 1 < nhid) ? l + 1 : l;
        const WSrc wn = make_wsrc(m.wp[ln] + (wave * NFT) * 256, lane);
        skew_layer<NFT, NKG>(acc, wh, bA, wl, wn, make_bias(m.bp[l], bias_voff), make_bias(m.bp[ln], bias_voff),
                             FTT * 64, bl, st_base, m.alpha ST_ARG);
        wl = wn;
    }
    // the last hidden layer's tile B (already activated)
#pragma unroll
    for (int idx = 0; idx < 4 * NFT; ++idx) store_quad<NFT>(acc, 1, idx, st_base);
}

// Output layer for NT feature tiles, K split over the four waves; partial sums go to
// part[(wave*out_dim + d)*MS + s] (floats, aliases the activation image).
template <int NT, int NST>
__device__ __forceinline__ void last_layer(const MlpArgs& m, int kq, char* act, float* part, int wave,
                                           int lane) {
    constexpr int MS = 32 * NST;
    const int h = lane >> 5, j = lane & 31;
    f32x16 acc[NT][NST];
    const WSrc wl = make_wsrc(m.wp[m.n_layers - 1] + (size_t)(wave * kq) * NT * 256, lane);
    f32x4 a0[NT], a1[NT], a2[NT];
    prefetch_a<NT>(a0, a1, a2, wl, NT * 64);
    gemm_tile<NT, NST>(acc, a0, a1, a2, wl, NT, kq, act + act_off<MS>(wave * kq, j, h), nullptr);
    __syncthreads();   // every wave is done reading the activation image
    const int od = m.out_dim;
#pragma unroll
    for (int ft = 0; ft < NT; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 32 * ft + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (d < od) {
#pragma unroll
                for (int st = 0; st < NST; ++st) part[(wave * od + d) * MS + 32 * st + j] = acc[ft][st][r];
            }
        }
    __syncthreads();
}

// Output layer when out_dim <= 4 (the 2-D sets: dy = 2; H for dx <= 4), 64-sample passes only.  A 32x32 tile
// would be 7/8 padding there (8192 cycles per pass and wave, 0.8 % of the bench workload).  v_mfma_f32_4x4x1_16B_f32
// runs 16 independent 4x4 blocks: block b = lanes 4b..4b+3, A_b[i][k] from lane 4b+i, B_b[k][j] from lane 4b+j,
// D_b[i][j] in register i of lane 4b+j.  With A_b[i] = W[k][i] for every block and B_b[j] = act[k][sample 4b+j], lane s
// ends with the four output features of sample s.  K is split over the waves like in last_layer; inside a quarter
// k ascends.  wsmall[((kg*2 + hh)*4 + i)*4 + e] = W[8kg + 4hh + e][i] (zero for i >= out_dim).
// small_load_w requests this wave's K-quarter of the weights (they do not depend on the activations: it is called
// before the barrier that publishes the last hidden layer, so the L2 round trip hides behind it).
template <int KQ>
__device__ __forceinline__ void small_load_w(const MlpArgs& m, int wave, int lane, f32x4 (&w)[2 * KQ]) {
    const f32x4* wp = (const f32x4*)m.wsmall + (size_t)(wave * KQ) * 8 + (lane & 3);
#pragma unroll
    for (int i = 0; i < 2 * KQ; ++i) w[i] = wp[i * 4];
    __builtin_amdgcn_sched_barrier(0);
}
template <int KQ>
__device__ __forceinline__ void last_layer_small(const MlpArgs& m, const f32x4 (&w)[2 * KQ], char* act, float* part,
                                                 int wave, int lane) {
    constexpr int MS = 64;
    f32x4 b[2 * KQ];
#pragma unroll
    for (int i = 0; i < 2 * KQ; ++i) b[i] = *(const f32x4*)(act + act_off<MS>(wave * KQ + (i >> 1), lane, i & 1));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2 * KQ; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i][e], b[i][e], acc, 0, 0, 0);
    __syncthreads();   // every wave is done reading the activation image
    const int od = m.out_dim;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d < od) part[(wave * od + d) * MS + lane] = acc[d];
    __syncthreads();
}

// Runs the MLP on the MS samples whose (padded) inputs sit in the x0 image.  On return
// part[(w*out_dim + d)*MS + s], w = 0..3, hold the four K-quarter partial sums of output d for sample s
// (bias not yet added).  Must be called by all 256 threads.
// pre: see mlp_prefetch (call it for the same pass before mlp_core, as early as the caller can).
template <int HID, int NST, bool JVP = false>
constexpr bool mlp_is_skewed() { return (NST == 2) && !JVP; }   // 64-sample passes: widths 128, 256, 512

template <int HID, int NST, bool JVP = false>
__device__ __forceinline__ void mlp_prefetch(const MlpArgs& m, MlpPre<HID>& pre, int wave, int lane) {
    if constexpr (mlp_is_skewed<HID, NST, JVP>()) skew_prefetch<HID>(m, pre, wave, lane);
}

template <int HID, int NST, bool JVP = false>
__device__ __forceinline__ void mlp_core(const MlpArgs& m, MlpPre<HID>& pre, char* x0, char* act, int wave,
                                         int lane ST_PARAM) {
    constexpr int NFT = HID / 128;
    constexpr int FTT = HID / 32;
    constexpr int MS = 32 * NST;
    const int h = lane >> 5, j = lane & 31;
    const int nhid = m.n_layers - 1;
    constexpr int KQ = HID / 32;        // k-groups per wave in the output layer
    const bool small_out = (NST == 2) && m.wsmall != nullptr;
    f32x4 wsm[2 * KQ];
    constexpr bool SKEWED = mlp_is_skewed<HID, NST, JVP>();
    if constexpr (SKEWED) {
        hidden_skewed<HID>(m, pre, x0, act, wave, lane ST_ARG);
        ST(1)
        if (small_out) small_load_w<KQ>(m, wave, lane, wsm);
        __syncthreads();
        ST(1)
    } else {
        f32x16 acc[NFT][NST];
        f32x16 bv[NFT];
        load_bias<NFT>(bv, m.bp[0], wave, lane);
        gemm_tile_simple<NFT, NST, JVP>(acc, make_wsrc(m.wp[0] + (wave * NFT) * 256, lane), FTT, m.k0 >> 3,
                                        x0 + act_off<MS>(0, j, h), bv);
        // hidden layers hand sample tile 0 over already activated (final_stage_act0); layer 0 does it here
        constexpr bool ACT0 = (NST == 2) && !JVP;
        if constexpr (ACT0) {
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
                for (int q = 0; q < 4; ++q) lrelu_quad(acc[ft][0], q, m.alpha);
        }
        ST(1)
        f32x4 a0[NFT], a1[NFT], a2[NFT];
        for (int l = 0; l + 1 < nhid; ++l) {
            const WSrc wnext = make_wsrc(m.wp[l + 1] + (wave * NFT) * 256, lane);
            prefetch_a<NFT>(a0, a1, a2, wnext, FTT * 64);
            if (l > 0) __syncthreads();   // all waves finished reading layer l's input
            ST(2)
            store_act<NFT, NST, JVP, ACT0>(acc, m.alpha, act, wave, lane);
            load_bias<NFT>(bv, m.bp[l + 1], wave, lane);
            ST(3)
            __syncthreads();
            ST(4)
            gemm_tile<NFT, NST, JVP, ACT0>(acc, a0, a1, a2, wnext, FTT, HID / 8, act + act_off<MS>(0, j, h), bv, m.alpha);
            ST(5)
        }
        // the last hidden layer's epilogue (kept out of the loop so the output layer's prefetched weights are not
        // live across the hidden layers' K loops)
        if (nhid > 1) __syncthreads();
        ST(2)
        store_act<NFT, NST, JVP, ACT0>(acc, m.alpha, act, wave, lane);
        if (small_out) small_load_w<KQ>(m, wave, lane, wsm);
        ST(3)
        __syncthreads();
        ST(4)
    }
    float* part = (float*)act;
    if (small_out) {
        last_layer_small<KQ>(m, wsm, act, part, wave, lane);
        ST(6)
        return;
    }
    switch (m.nt_out) {
        case 1: last_layer<1, NST>(m, HID / 32, act, part, wave, lane); break;
        case 2: last_layer<2, NST>(m, HID / 32, act, part, wave, lane); break;
        case 3: last_layer<3, NST>(m, HID / 32, act, part, wave, lane); break;
        default: last_layer<4, NST>(m, HID / 32, act, part, wave, lane); break;
    }
    ST(6)
}

template <int MS>
__device__ __forceinline__ float mlp_out_nobias(const MlpArgs& m, const float* part, int d, int s) {
    const int od = m.out_dim;
    const float p0 = part[(0 * od + d) * MS + s], p1 = part[(1 * od + d) * MS + s];
    const float p2 = part[(2 * od + d) * MS + s], p3 = part[(3 * od + d) * MS + s];
    return ((p0 + p1) + p2) + p3;
}
template <int MS>
__device__ __forceinline__ float mlp_out(const MlpArgs& m, const float* part, int d, int s) {
    return mlp_out_nobias<MS>(m, part, d, s) + m.bp[m.n_layers - 1][d];
}

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

__device__ __forceinline__ float u01(uint32_t x) {   // (0,1), 24 bits, never 0 or 1
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = sqrtf(-2.0f * logf(u01(a)));
    float sn, cs;
    sincosf(6.28318530717958647692f * u01(b), &sn, &cs);
    n0 = r * cs;
    n1 = r * sn;
}

struct SamplerArgs {
    uint32_t seed_lo, seed_hi;
    int ndf;        // df if it is an integer in [1, 32], else 0 (gamma path)
    float df;
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

// Calls put(d, x_d) for d = 0 .. dx-1 with the proposal sample (yidx, i) around xhat.
template <typename Put>
__device__ __forceinline__ void draw_sample(const SamplerArgs& sa, uint64_t yidx, uint32_t i,
                                            const float* xhat, int dx, Put put) {
    const uint32_t ylo = (uint32_t)yidx, yhi = (uint32_t)(yidx >> 32);
    float zsum = 0.0f, mult = 0.0f;
    if (sa.ndf == 0) mult = sa.scale / sqrtf(chi2_over_df_gamma(sa, i, ylo, yhi));
    const int total = sa.ndf + dx;
    for (int c = 0; 4 * c < total; ++c) {
        const U4 r = philox4x32_10(i, (uint32_t)c, ylo, yhi, sa.seed_lo, sa.seed_hi);
        float n[4];
        box_muller(r.x, r.y, n[0], n[1]);
        box_muller(r.z, r.w, n[2], n[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = 4 * c + k;
            if (g < sa.ndf) {
                zsum += n[k] * n[k];
                if (g == sa.ndf - 1) mult = sa.scale / sqrtf(zsum / sa.df);
            } else if (g < total) {
                const int d = g - sa.ndf;
                put(d, xhat[d] + mult * n[k]);
            }
        }
    }
}

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

// out[k*N + n] for sd index k: merges the chunks of query point n.
__global__ void is_finalize_kernel(const double* partial, int64_t N, int nchunks, int n_sd, double logS,
                                   int log_flag, double* out) {
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
        out[(int64_t)k * N + n] = log_flag ? lp : exp(lp);
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

// ----------------------------------------------------------------------------------------------
// Host side
// ----------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct HostMlp {
    bool set = false;
    int n_layers = 0;
    int hid = 0;              // hidden width the kernels run at: max hidden width padded up to 128/256/512/1024
    int nst = 2;              // sample tiles per wave (2: 64 samples per pass; 1: 32, for width 1024)
    std::vector<int> dims;
    std::vector<void*> dev;   // device allocations to free
    MlpArgs args;
    double flops_per_row = 0.0;
};

}  // namespace

struct rt_ctx {
    int device = 0;
    int cu_count = 0;
    hipStream_t stream = nullptr;
    HostMlp mlp[2];
    DevBuf b_y, b_xhat, b_xs, b_out, b_partial, b_in, b_fout;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double flops_acc = 0.0;
    uint64_t evals_acc = 0;
    std::string err;
};

namespace {

std::string g_create_err;

int fail(rt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, RT_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

int ensure(rt_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return RT_OK;
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t cap = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipMalloc(&b.p, cap));
    b.cap = cap;
    return RT_OK;
}

// Packs W [K x NF] (row-major, out = in @ W), zero-padded to [Kp x NFp], into per-lane MFMA A-operand order:
//   packed[((kg*FTT + ft)*64 + lane)*4 + e] = W[8kg + 4h + e][32ft + i],  lane = 32h + i,
// zero where k >= K or the feature >= NF.  FTT = ceil(NFp/32), kg < ceil(Kp/8).
// Padding hidden widths with zero weights and biases is exact: a padded unit is lrelu(0) = 0 and feeds 0*w.
void pack_layer(const float* W, int K, int NF, int Kp, int NFp, std::vector<float>& out) {
    const int nkg = (Kp + 7) / 8, ftt = (NFp + 31) / 32;
    out.assign((size_t)nkg * ftt * 64 * 4, 0.0f);
    for (int kg = 0; kg < nkg; ++kg)
        for (int ft = 0; ft < ftt; ++ft)
            for (int lane = 0; lane < 64; ++lane) {
                const int h = lane >> 5, i = lane & 31, f = 32 * ft + i;
                if (f >= NF) continue;
                for (int e = 0; e < 4; ++e) {
                    const int k = 8 * kg + 4 * h + e;
                    if (k < K) out[(((size_t)kg * ftt + ft) * 64 + lane) * 4 + e] = W[(size_t)k * NF + f];
                }
            }
}

void free_mlp(HostMlp& m) {
    for (void* p : m.dev) (void)hipFree(p);
    m.dev.clear();
    m.set = false;
}

template <int HID, int NST>
int launch_is(rt_ctx* ctx, const IsArgs& a, bool given, int grid, int lds) {
    if (given) {
        HIPCHK(ctx, hipFuncSetAttribute((const void*)is_fused_kernel<HID, NST, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((is_fused_kernel<HID, NST, true>), dim3(grid), dim3(NTHREADS), lds, ctx->stream, a);
    } else {
        HIPCHK(ctx, hipFuncSetAttribute((const void*)is_fused_kernel<HID, NST, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((is_fused_kernel<HID, NST, false>), dim3(grid), dim3(NTHREADS), lds, ctx->stream, a);
    }
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

template <int HID, int NST>
int launch_fwd(rt_ctx* ctx, const MlpArgs& m, const float* in, int64_t N, float* out, int grid, int lds) {
    HIPCHK(ctx, hipFuncSetAttribute((const void*)mlp_fwd_kernel<HID, NST>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((mlp_fwd_kernel<HID, NST>), dim3(grid), dim3(NTHREADS), lds, ctx->stream, m, in, N, out);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

template <int HID>
int launch_jvp(rt_ctx* ctx, const MlpArgs& m, const float* in, int64_t N, float* yout, float* J, int grid, int lds) {
    HIPCHK(ctx, hipFuncSetAttribute((const void*)mlp_jvp_kernel<HID>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((mlp_jvp_kernel<HID>), dim3(grid), dim3(NTHREADS), lds, ctx->stream, m, in, N, yout, J);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

int make_sampler(rt_ctx* ctx, double scale, double df, uint64_t seed, SamplerArgs& sa) {
    if (!(scale > 0.0) || !(df > 0.0) || !isfinite(scale) || !isfinite(df))
        return fail(ctx, RT_EINVAL, "scale and df must be positive and finite");
    sa.seed_lo = (uint32_t)seed;
    sa.seed_hi = (uint32_t)(seed >> 32);
    sa.df = (float)df;
    sa.scale = (float)scale;
    sa.ndf = (df == floor(df) && df <= 32.0) ? (int)df : 0;
    return RT_OK;
}

// Enqueues the fused kernel + finalize.  All pointers are device pointers.
int is_enqueue(rt_ctx* ctx, const double* y, const float* xhat, const double* xs, int64_t N,
               int64_t y_index_base, int64_t S, int n_sd, const double* sd_ys, double scale, double df,
               uint64_t seed, int log_flag, double* out) {
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set");
    if (N < 0 || S <= 0 || S > 0xFFFFFFFFll) return fail(ctx, RT_EINVAL, "need N >= 0 and 0 < S < 2^32");
    if (n_sd < 1 || n_sd > RT_MAX_SD || !sd_ys) return fail(ctx, RT_EINVAL, "n_sd must be in [1, RT_MAX_SD]");
    for (int k = 0; k < n_sd; ++k)
        if (!(sd_ys[k] > 0.0) || !isfinite(sd_ys[k])) return fail(ctx, RT_EINVAL, "sd_y must be positive and finite");
    if (N == 0) return RT_OK;
    IsArgs a;
    memset(&a, 0, sizeof(a));
    a.g = G.args;
    int rc = make_sampler(ctx, scale, df, seed, a.sa);
    if (rc) return rc;
    const int dx = G.args.in_dim, dy = G.args.out_dim;
    // work units: (query point, chunk of samples); aim for >= 128 units per CU-resident workgroup
    const int nst = G.nst;
    const int MS = 32 * nst;
    const int grid_max = ctx->cu_count;
    const int64_t passes = (S + MS - 1) / MS;
    int64_t want = (128ll * grid_max + N - 1) / N;
    if (want < 1) want = 1;
    if (want > passes) want = passes;
    const int64_t nchunks = want;
    a.nchunks = (int)nchunks;
    a.passes = passes;
    a.n_units = N * nchunks;
    a.N = N; a.S = S; a.y_index_base = y_index_base;
    a.y = y; a.xhat = xhat; a.xs = xs;
    a.n_sd = n_sd;
    rc = ensure(ctx, ctx->b_partial, (size_t)a.n_units * n_sd * 2 * sizeof(double));
    if (rc) return rc;
    a.partial = (double*)ctx->b_partial.p;
    const double PI = 3.14159265358979323846;
    const double ct = lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df * PI) - log(scale);
    for (int k = 0; k < n_sd; ++k) {
        const double sd_y = sd_ys[k];
        a.c_all[k] = -(double)dy * log(sqrt(2.0 * PI) * sd_y) - (double)dx * log(sqrt(2.0 * PI)) - (double)dx * ct;
        a.inv_2sd2[k] = 1.0 / (2.0 * sd_y * sd_y);
    }
    a.inv_scale = 1.0 / scale;
    a.inv_df = 1.0 / df;
    a.half_dfp1 = 0.5 * (df + 1.0);

    const Carve cv = carve_for(G.hid, MS, G.args.k0, dx, dy);
    const int grid = (int)((a.n_units < grid_max) ? a.n_units : grid_max);
    if (ctx->events_used == ctx->events.size()) {
        hipEvent_t e0, e1;
        HIPCHK(ctx, hipEventCreate(&e0));
        HIPCHK(ctx, hipEventCreate(&e1));
        ctx->events.emplace_back(e0, e1);
    }
    auto& ev = ctx->events[ctx->events_used];
    HIPCHK(ctx, hipEventRecord(ev.first, ctx->stream));
    const bool given = xs != nullptr;
    switch (G.hid * 10 + nst) {
        case 1282: rc = launch_is<128, 2>(ctx, a, given, grid, cv.total); break;
        case 2562: rc = launch_is<256, 2>(ctx, a, given, grid, cv.total); break;
        case 5122: rc = launch_is<512, 2>(ctx, a, given, grid, cv.total); break;
        case 10241: rc = launch_is<1024, 1>(ctx, a, given, grid, cv.total); break;
        default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
    }
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(ev.second, ctx->stream));
    ctx->events_used++;
    ctx->flops_acc += (double)N * (double)S * G.flops_per_row;
    ctx->evals_acc += (uint64_t)N * (uint64_t)S;
    const int fb = 256;
    hipLaunchKernelGGL(is_finalize_kernel, dim3((unsigned)((N + fb - 1) / fb)), dim3(fb), 0, ctx->stream,
                       (const double*)a.partial, N, a.nchunks, n_sd, log((double)S), log_flag, out);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

int fwd_enqueue(rt_ctx* ctx, int which, const float* in, int64_t N, float* out) {
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    HostMlp& M = ctx->mlp[which];
    if (!M.set) return fail(ctx, RT_ENOTSET, "MLP has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    const int ms = 32 * M.nst;
    const Carve cv = carve_for(M.hid, ms, M.args.k0, M.args.in_dim, M.args.out_dim);
    const int64_t npass = (N + ms - 1) / ms;
    const int64_t gmax = (int64_t)ctx->cu_count * 2;
    const int grid = (int)(npass < gmax ? npass : gmax);
    switch (M.hid) {
        case 128: return launch_fwd<128, 2>(ctx, M.args, in, N, out, grid, cv.total);
        case 256: return launch_fwd<256, 2>(ctx, M.args, in, N, out, grid, cv.total);
        case 512: return launch_fwd<512, 2>(ctx, M.args, in, N, out, grid, cv.total);
        case 1024: return launch_fwd<1024, 1>(ctx, M.args, in, N, out, grid, cv.total);
        default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
    }
}

int jvp_enqueue(rt_ctx* ctx, int which, const float* in, int64_t N, float* yout, float* J) {
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    HostMlp& M = ctx->mlp[which];
    if (!M.set) return fail(ctx, RT_ENOTSET, "MLP has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    if (M.nst != 2) return fail(ctx, RT_EUNSUPPORTED, "Jacobian needs hidden width <= 512 (64-column passes)");
    const Carve cv = carve_for(M.hid, 64, M.args.k0, M.args.in_dim, M.args.out_dim);
    const int64_t npass = ((N + 31) / 32) * M.args.in_dim;
    const int64_t gmax = (int64_t)ctx->cu_count * 2;
    const int grid = (int)(npass < gmax ? npass : gmax);
    switch (M.hid) {
        case 128: return launch_jvp<128>(ctx, M.args, in, N, yout, J, grid, cv.total);
        case 256: return launch_jvp<256>(ctx, M.args, in, N, yout, J, grid, cv.total);
        case 512: return launch_jvp<512>(ctx, M.args, in, N, yout, J, grid, cv.total);
        default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
    }
}

}  // namespace

extern "C" {

int rt_ctx_create(int device, rt_ctx** out) {
    if (!out) return fail(nullptr, RT_EINVAL, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    HIPCHK(nullptr, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(nullptr, RT_EINVAL, "no such HIP device");
    HIPCHK(nullptr, hipSetDevice(device));
    rt_ctx* ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_ENOMEM, "out of host memory");
    ctx->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, RT_EHIP, hipGetErrorString(e)); }
    ctx->cu_count = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, RT_EHIP, hipGetErrorString(e)); }
    *out = ctx;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_mlp(ctx->mlp[0]);
    free_mlp(ctx->mlp[1]);
    DevBuf* bufs[] = {&ctx->b_y, &ctx->b_xhat, &ctx->b_xs, &ctx->b_out, &ctx->b_partial, &ctx->b_in, &ctx->b_fout};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int rt_set_mlp(rt_ctx* ctx, int which, int n_layers, const int32_t* dims, const float* const* W,
               const float* const* b, float lrelu_alpha) {
    if (!ctx) return RT_EINVAL;
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    if (!dims || !W || !b) return fail(ctx, RT_EINVAL, "NULL argument");
    if (n_layers < 2 || n_layers > RT_MAX_LAYERS) return fail(ctx, RT_EINVAL, "n_layers must be in [2, RT_MAX_LAYERS]");
    if (!(lrelu_alpha >= 0.0f && lrelu_alpha <= 1.0f)) return fail(ctx, RT_EINVAL, "lrelu_alpha must be in [0, 1]");
    for (int l = 0; l <= n_layers; ++l) if (dims[l] < 1) return fail(ctx, RT_EINVAL, "dims must be >= 1");
    for (int l = 0; l < n_layers; ++l) if (!W[l] || !b[l]) return fail(ctx, RT_EINVAL, "NULL weight or bias");
    int hmax = 0;
    for (int l = 1; l < n_layers; ++l) if (dims[l] > hmax) hmax = dims[l];
    if (hmax > 1024) return fail(ctx, RT_EUNSUPPORTED, "hidden width > 1024");
    // the kernels run at one of four widths; narrower / unequal hidden layers are zero-padded up (exact)
    const int hid = hmax <= 128 ? 128 : hmax <= 256 ? 256 : hmax <= 512 ? 512 : 1024;
    const int nst = hid == 1024 ? 1 : 2;
    if (dims[0] > MAX_IN) return fail(ctx, RT_EUNSUPPORTED, "input dim > 128");
    if (dims[n_layers] > MAX_OUT) return fail(ctx, RT_EUNSUPPORTED, "output dim > 128");
    if (carve_for(hid, 32 * nst, ((dims[0] + 7) / 8) * 8, dims[0], dims[n_layers]).total > LDS_MAX)
        return fail(ctx, RT_EUNSUPPORTED, "input dim too large for this hidden width (LDS budget of 160 KiB)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HostMlp& M = ctx->mlp[which];
    free_mlp(M);
    M.n_layers = n_layers;
    M.hid = hid;
    M.nst = nst;
    M.dims.assign(dims, dims + n_layers + 1);
    memset(&M.args, 0, sizeof(M.args));
    M.args.n_layers = n_layers;
    M.args.in_dim = dims[0];
    M.args.out_dim = dims[n_layers];
    M.args.k0 = ((dims[0] + 7) / 8) * 8;
    M.args.nt_out = (dims[n_layers] + 31) / 32;
    M.args.alpha = lrelu_alpha;
    M.flops_per_row = 0.0;
    std::vector<float> packed, bias;
    for (int l = 0; l < n_layers; ++l) {
        const int K = dims[l], NF = dims[l + 1];
        const int Kp = (l == 0) ? K : hid, NFp = (l == n_layers - 1) ? NF : hid;
        M.flops_per_row += 2.0 * (double)K * (double)NF;   // algorithmic (unpadded) flops
        pack_layer(W[l], K, NF, Kp, NFp, packed);
        // one extra zero k-group of padding keeps any speculative prefetch in bounds
        const size_t pad = (size_t)((NFp + 31) / 32) * 64 * 4;
        packed.resize(packed.size() + pad, 0.0f);
        void* dw = nullptr;
        HIPCHK(ctx, hipMalloc(&dw, packed.size() * sizeof(float)));
        M.dev.push_back(dw);
        HIPCHK(ctx, hipMemcpy(dw, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        bias.assign((size_t)((NFp + 31) / 32) * 32, 0.0f);
        memcpy(bias.data(), b[l], sizeof(float) * NF);
        void* db = nullptr;
        HIPCHK(ctx, hipMalloc(&db, bias.size() * sizeof(float)));
        M.dev.push_back(db);
        HIPCHK(ctx, hipMemcpy(db, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
        M.args.wp[l] = (const float*)dw;
        M.args.bp[l] = (const float*)db;
    }
    if (nst == 2 && dims[n_layers] <= 4) {
        // output layer in last_layer_small's order: wsmall[((kg*2 + hh)*4 + i)*4 + e] = W[8kg + 4hh + e][i]
        const int l = n_layers - 1, K = dims[l], NF = dims[l + 1];
        std::vector<float> ws((size_t)hid * 4, 0.0f);
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < NF; ++i)
                ws[(size_t)(((k >> 3) * 2 + ((k >> 2) & 1)) * 4 + i) * 4 + (k & 3)] = W[l][(size_t)k * NF + i];
        void* dws = nullptr;
        HIPCHK(ctx, hipMalloc(&dws, ws.size() * sizeof(float)));
        M.dev.push_back(dws);
        HIPCHK(ctx, hipMemcpy(dws, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice));
        M.args.wsmall = (const float*)dws;
    }
    M.set = true;
    return RT_OK;
}

int rt_mlp_forward_dev(rt_ctx* ctx, int which, const float* in_dev, int64_t N, float* out_dev) {
    if (!ctx) return RT_EINVAL;
    if (N > 0 && (!in_dev || !out_dev)) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return fwd_enqueue(ctx, which, in_dev, N, out_dev);
}

int rt_mlp_forward(rt_ctx* ctx, int which, const float* in, int64_t N, float* out) {
    if (!ctx) return RT_EINVAL;
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    HostMlp& M = ctx->mlp[which];
    if (!M.set) return fail(ctx, RT_ENOTSET, "MLP has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    if (!in || !out) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t ib = (size_t)N * M.args.in_dim * sizeof(float), ob = (size_t)N * M.args.out_dim * sizeof(float);
    int rc = ensure(ctx, ctx->b_in, ib); if (rc) return rc;
    rc = ensure(ctx, ctx->b_fout, ob); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_in.p, in, ib, hipMemcpyHostToDevice, ctx->stream));
    rc = fwd_enqueue(ctx, which, (const float*)ctx->b_in.p, N, (float*)ctx->b_fout.p);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_fout.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_mlp_jacobian(rt_ctx* ctx, int which, const float* in, int64_t N, float* out, float* J) {
    if (!ctx) return RT_EINVAL;
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    HostMlp& M = ctx->mlp[which];
    if (!M.set) return fail(ctx, RT_ENOTSET, "MLP has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    if (!in || !J) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t id = M.args.in_dim, od = M.args.out_dim;
    const size_t ib = (size_t)N * id * sizeof(float), ob = (size_t)N * od * sizeof(float), jb = ob * id;
    int rc = ensure(ctx, ctx->b_in, ib); if (rc) return rc;
    rc = ensure(ctx, ctx->b_fout, ob + jb); if (rc) return rc;
    float* d_out = (float*)ctx->b_fout.p;
    float* d_J = d_out + (size_t)N * od;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_in.p, in, ib, hipMemcpyHostToDevice, ctx->stream));
    rc = jvp_enqueue(ctx, which, (const float*)ctx->b_in.p, N, d_out, d_J);
    if (rc) return rc;
    if (out) HIPCHK(ctx, hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(J, d_J, jb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_encode(rt_ctx* ctx, const float* y, int64_t N, float* xhat) {
    return rt_mlp_forward(ctx, RT_MLP_H, y, N, xhat);
}

static int is_host(rt_ctx* ctx, const double* y, const float* xhat, const double* xs, int64_t N,
                   int64_t y_index_base, int64_t S, int n_sd, const double* sd_ys, double scale, double df,
                   uint64_t seed, int log_flag, double* out, bool given) {
    if (!ctx) return RT_EINVAL;
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    if (!y || !xhat || !out || (given && !xs) || !sd_ys) return fail(ctx, RT_EINVAL, "NULL argument");
    if (S <= 0) return fail(ctx, RT_EINVAL, "S must be positive");
    if (n_sd < 1 || n_sd > RT_MAX_SD) return fail(ctx, RT_EINVAL, "n_sd must be in [1, RT_MAX_SD]");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int dx = G.args.in_dim, dy = G.args.out_dim;
    const size_t yb = (size_t)N * dy * sizeof(double), xb = (size_t)N * dx * sizeof(float);
    int rc = ensure(ctx, ctx->b_y, yb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_xhat, xb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_out, (size_t)N * n_sd * sizeof(double)); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_y.p, y, yb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_xhat.p, xhat, xb, hipMemcpyHostToDevice, ctx->stream));
    const double* xs_dev = nullptr;
    if (given) {
        const size_t sbytes = (size_t)N * (size_t)S * dx * sizeof(double);
        rc = ensure(ctx, ctx->b_xs, sbytes); if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->b_xs.p, xs, sbytes, hipMemcpyHostToDevice, ctx->stream));
        xs_dev = (const double*)ctx->b_xs.p;
    }
    rc = is_enqueue(ctx, (const double*)ctx->b_y.p, (const float*)ctx->b_xhat.p, xs_dev, N, y_index_base,
                    S, n_sd, sd_ys, scale, df, seed, log_flag, (double*)ctx->b_out.p);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_out.p, (size_t)N * n_sd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_is_estimate(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base,
                   int64_t S, double sd_y, double scale, double df, uint64_t seed, int log_flag,
                   double* out) {
    return is_host(ctx, y, xhat, nullptr, N, y_index_base, S, 1, &sd_y, scale, df, seed, log_flag, out, false);
}

int rt_is_estimate_multi(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base,
                         int64_t S, int n_sd, const double* sd_y, double scale, double df, uint64_t seed,
                         int log_flag, double* out) {
    return is_host(ctx, y, xhat, nullptr, N, y_index_base, S, n_sd, sd_y, scale, df, seed, log_flag, out, false);
}

int rt_is_estimate_given_samples(rt_ctx* ctx, const double* y, const float* xhat, const double* x_samples,
                                 int64_t N, int64_t S, double sd_y, double scale, double df, int log_flag,
                                 double* out) {
    return is_host(ctx, y, xhat, x_samples, N, 0, S, 1, &sd_y, scale, df, 0, log_flag, out, true);
}

int rt_is_estimate_dev(rt_ctx* ctx, const double* y_dev, const float* xhat_dev, int64_t N,
                       int64_t y_index_base, int64_t S, double sd_y, double scale, double df,
                       uint64_t seed, int log_flag, double* out_dev) {
    if (!ctx) return RT_EINVAL;
    if (N > 0 && (!y_dev || !xhat_dev || !out_dev)) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return is_enqueue(ctx, y_dev, xhat_dev, nullptr, N, y_index_base, S, 1, &sd_y, scale, df, seed, log_flag, out_dev);
}

int rt_draw_samples(rt_ctx* ctx, const float* xhat, int64_t N, int64_t y_index_base, int64_t S,
                    double scale, double df, uint64_t seed, float* x_samples) {
    if (!ctx) return RT_EINVAL;
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set (it defines dx)");
    if (N < 0 || S <= 0 || S > 0xFFFFFFFFll) return fail(ctx, RT_EINVAL, "need N >= 0 and 0 < S < 2^32");
    if (N == 0) return RT_OK;
    if (!xhat || !x_samples) return fail(ctx, RT_EINVAL, "NULL argument");
    SamplerArgs sa;
    int rc = make_sampler(ctx, scale, df, seed, sa);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int dx = G.args.in_dim;
    const size_t xb = (size_t)N * dx * sizeof(float), sb = (size_t)N * (size_t)S * dx * sizeof(float);
    rc = ensure(ctx, ctx->b_xhat, xb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_xs, sb); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_xhat.p, xhat, xb, hipMemcpyHostToDevice, ctx->stream));
    const int64_t total = N * S;
    hipLaunchKernelGGL(draw_samples_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       sa, (const float*)ctx->b_xhat.p, N, y_index_base, S, dx, (float*)ctx->b_xs.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(x_samples, ctx->b_xs.p, sb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_sync(rt_ctx* ctx) {
    if (!ctx) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_get_counters(rt_ctx* ctx, rt_counters* out) {
    if (!ctx || !out) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float t = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&t, ctx->events[i].first, ctx->events[i].second));
        ms += t;
    }
    out->kernel_ms = ms;
    out->flops = ctx->flops_acc;
    out->launches = ctx->events_used;
    out->sample_evals = ctx->evals_acc;
    return RT_OK;
}

int rt_reset_counters(rt_ctx* ctx) {
    if (!ctx) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->events_used = 0;
    ctx->flops_acc = 0.0;
    ctx->evals_acc = 0;
    return RT_OK;
}

#ifdef RT_STAMPS
int rt_debug_stamps(rt_ctx* ctx, unsigned long long* out48, int reset) {
    if (!ctx || !out48) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 4 * NSTAMP));
    if (reset) {
        unsigned long long z[4 * NSTAMP] = {0};
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)));
    }
    return RT_OK;
}
#endif

int rt_device_cu_count(rt_ctx* ctx, int* out) {
    if (!ctx || !out) return RT_EINVAL;
    *out = ctx->cu_count;
    return RT_OK;
}

}  // extern "C"
