// rt_mlp.hpp — the fp32-MFMA multilayer perceptron of one workgroup pass: LDS activation image, packed-weight
// streaming, the plain and the skewed layer schedules, output layers.  Included by rt_is.hip only (one translation
// unit; everything lives in its anonymous namespace).  Design and numbers: DESIGN.md §4.
#pragma once


constexpr int NTHREADS = 256;   // 4 waves, one per SIMD (NW = 4, the default of every template below)
// NW = waves per workgroup.  NW = 4: a workgroup per CU, the hidden layer's features split over its four waves, barriers
// between the layer stages.  NW = 1 (hidden width <= 128, see rt_set_mlp): a workgroup is ONE wave that owns all features
// of its own 64 samples — four of them share a CU, one per SIMD; there is nothing to synchronise with, so hipcc drops the
// s_barrier of every __syncthreads() (the LDS fence stays) and the per-pass sampler / scorer / reduction, which run on
// one wave per workgroup, keep all four SIMDs busy instead of one.  NW = 2 (width 256): two waves with 128 features each,
// two workgroups to a CU — one SIMD in two idles during a workgroup's serial phases instead of three in four.
// Same arithmetic in the same order in every case: bit-identical results.
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
__device__ unsigned long long g_wg_ticks[2 * 1024];   // per workgroup: ticks from kernel start to its end, and its start stamp
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
template <int NFT, int NST, bool FIRST = false>
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
                    (FIRST && e == 0) ? (cinit ? cinit[ft] : zero) : acc[ft][st], 0, 0, 0);
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
template <int NFT, int NST>
__device__ __forceinline__ void gemm_tile_simple(f32x16 (&acc)[NFT][NST], const WSrc& wl,
                                                 int ftt, int nkg, const char* bl, const f32x16* cinit) {
    f32x4 a[NFT], b[NST];
    load_group<NFT, NST>(a, b, wl, ftt * 64, bl, 0);
    mfma_group<NFT, NST, true>(acc, a, b, cinit);
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
template <int NFT, int NST, bool FIRST = false>
__device__ __forceinline__ void pipe_stage(f32x16 (&acc)[NFT][NST], f32x4 (&an)[NFT], f32x4 (&bn)[NST],
                                           const f32x4 (&ac)[NFT], const f32x4 (&bc)[NST],
                                           const WSrc& wl, int wstride, const char* bl, int kgn,
                                           const f32x16* cinit = nullptr) {
    load_group<NFT, NST>(an, bn, wl, wstride, bl, kgn);
    mfma_group<NFT, NST, FIRST>(acc, ac, bc, cinit);
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

// nkg must be a positive multiple of 4.  Operands of k-group kg+3 are requested while k-group kg's
// MFMAs issue.  Tail prefetches are clamped to the last group (redundant, in bounds).
// a0..a2 arrive pre-requested (prefetch_a); only the LDS operands of groups 0..2 are read here.
// ACT0: on return sample tile 0's accumulators already hold lrelu(pre-activation) (see final_stage_act0).
template <int NFT, int NST, bool ACT0 = false>
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
    pipe_stage<NFT, NST, true>(acc, a3, b3, a0, b0, wl, wstride, bl, 3, cinit);
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
// ACT0: sample tile 0 is already activated (gemm_tile<..., ACT0>); only tile 1's arithmetic is left.
template <int NFT, int NST, bool ACT0 = false>
__device__ __forceinline__ void store_act(f32x16 (&acc)[NFT][NST], float alpha, char* act, int wave, int lane) {
    constexpr int MS = 32 * NST;
    const int h = lane >> 5, j = lane & 31;
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
// so a slot is a fixed register and any NKG >= 8 works: widths 64 (one-wave workgroups), 128, 256 and 512.
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

// The tangent passes' counterpart of lrelu_tile (mlp_jvp_rest_kernel): the tile holds tangent pre-activations; element r of
// feature tile ft keeps slope 1 where bit 16 ft + r of `mask` — the sign of the PRIMAL pre-activation in the same lane and
// register, left by the rows' first pass (store_act) — is set, alpha elsewhere.  One VALU cluster, like lrelu_tile.
template <int NFT, int TILE>
__device__ __forceinline__ void mask_tile(f32x16 (&acc)[NFT][2], float alpha, unsigned long long mask) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[ft][TILE][r];
            acc[ft][TILE][r] = ((mask >> (16 * ft + r)) & 1ull) ? d : d * alpha;
        }
    __builtin_amdgcn_sched_barrier(0);
}

// lrelu_tile for the PRIMAL tile of a Jacobian pass (tile 0), which also returns the mask of its positive pre-activations
// for the tangent tile(s) of the same rows (mask_tile).
template <int NFT>
__device__ __forceinline__ unsigned long long lrelu_mask_tile(f32x16 (&acc)[NFT][2], float alpha) {
    static_assert(NFT <= 4, "64 mask bits per thread and layer");
    unsigned long long mask = 0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = acc[ft][0][r];
            if (t > 0.0f) mask |= 1ull << (16 * ft + r);
            acc[ft][0][r] = vmax(t, t * alpha);
        }
    __builtin_amdgcn_sched_barrier(0);
    return mask;
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
// JM (the Jacobian kernels; mkp = this thread's slot for this layer's slope mask): 1 — tile A carries primal rows, tile B
// tangent columns of the same rows: A's cluster is lrelu_mask_tile (the mask goes to *mkp when mkp is not null), B's is
// mask_tile with it; `bias` (tile B's) is then an all-zero vector.  2 — both tiles carry tangent columns, of rows whose
// primal left the mask at *mkp (tile A's rows) and *mkpB (tile B's: the same rows or another block's): both clusters
// are mask_tile.  0: the forward pass as it always was.
template <int NFT, int NKG, int JM = 0>
__device__ __forceinline__ void skew_layer(f32x16 (&acc)[NFT][2], f32x4 (&wh)[3][NFT], f32x4 (&bA)[6],
                                           const WSrc& wl, const WSrc& wn, const BiasSrc& bias, const BiasSrc& bias_n,
                                           int wstride, const LdsBase& bl, char* st_base, float alpha ST_PARAM,
                                           unsigned long long* mkp = nullptr,
                                           const unsigned long long* mkpB = nullptr) {
    unsigned long long mk = 0, mkB = 0;
    if constexpr (JM == 2) { mk = *mkp; mkB = *mkpB; }
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
    if constexpr (JM == 1) { mk = lrelu_mask_tile<NFT>(acc, alpha); if (mkp) *mkp = mk; }
    else if constexpr (JM == 2) mask_tile<NFT, 0>(acc, alpha, mk);
    else lrelu_tile<NFT, 0>(acc, alpha);
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
    if constexpr (JM == 2) mask_tile<NFT, 1>(acc, alpha, mkB);
    else if constexpr (JM == 1) mask_tile<NFT, 1>(acc, alpha, mk);
    else lrelu_tile<NFT, 1>(acc, alpha);
    ST(13)
}

// Operands of a pass that do not depend on its samples — layer 0's bias (straight into both tiles' accumulators) and
// its first weights — requested BEFORE the pass's samples are drawn, so that their round trip (they have been evicted
// from L2 by the 9 MB of weights streamed since the previous pass) hides behind the sampler.
template <int HID, int NW = 4>
struct MlpPre {
    f32x16 acc[HID / (32 * NW)][2];
    f32x4 a[HID / (32 * NW)];
};
// bias_b (Jacobian passes whose tile 1 carries tangent columns): tile 1's bias vector — all zeros — instead of layer 0's.
template <int HID, int NW = 4>
__device__ __forceinline__ void skew_prefetch(const MlpArgs& m, MlpPre<HID, NW>& pre, int wave, int lane,
                                              const float* bias_b = nullptr) {
    constexpr int NFT = HID / (32 * NW), FTT = HID / 32;
    const BiasSrc b0 = make_bias(m.bp[0], (32 * wave * NFT + 4 * (lane >> 5)) * 4);
    const BiasSrc b0b = make_bias(bias_b ? bias_b : m.bp[0], (32 * wave * NFT + 4 * (lane >> 5)) * 4);
    const WSrc w0 = make_wsrc(m.wp[0] + (wave * NFT) * 256, lane);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) pre.a[ft] = wload(w0, 0, FTT * 64, ft);
#pragma unroll
    for (int idx = 0; idx < 4 * NFT; ++idx) { bias_quad<NFT, 0>(pre.acc, b0, idx); bias_quad<NFT, 1>(pre.acc, b0b, idx); }
    __builtin_amdgcn_sched_barrier(0);
}

// Layer 0 and all hidden layers after it.  On return the last hidden layer's activations are in the image, the
// closing barrier still to come (the caller requests the output layer's weights first).
// Layer 0 (K = k0, a few k-groups) is the same dance without the skew: all k-groups but the last for both tiles, then
// the last one: tile A's MFMAs, A's leaky-ReLU cluster, tile B's MFMAs with A's stores behind them, barrier, B's cluster.
// JM: see skew_layer; mask_ws = the rows' slope masks, [hidden layer][thread] (JM = 1: written, may be null; JM = 2: read,
// tile A's rows; mask_wsB: tile B's rows); zero_bias (JM = 1): the all-zero bias vector of the tangent tile.
template <int HID, int NW = 4, int JM = 0>
__device__ __forceinline__ void hidden_skewed(const MlpArgs& m, MlpPre<HID, NW>& pre, const char* x0, char* act, int wave,
                                              int lane ST_PARAM, unsigned long long* mask_ws = nullptr,
                                              const float* zero_bias = nullptr,
                                              const unsigned long long* mask_wsB = nullptr) {
    constexpr int NFT = HID / (32 * NW), FTT = HID / 32, NKG = HID / 8;
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
    unsigned long long mk0 = 0;       // layer 0's slope mask (Jacobian passes only)
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
        if constexpr (JM == 1) { mk0 = lrelu_mask_tile<NFT>(acc, m.alpha); if (mask_ws) mask_ws[wave * 64 + lane] = mk0; }
        else if constexpr (JM == 2) {
            mk0 = mask_ws[wave * 64 + lane];
            mask_tile<NFT, 0>(acc, m.alpha, mk0);
            mk0 = mask_wsB[wave * 64 + lane];
        }
        else lrelu_tile<NFT, 0>(acc, m.alpha);
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
    if constexpr (JM != 0) mask_tile<NFT, 1>(acc, m.alpha, mk0);
    else lrelu_tile<NFT, 1>(acc, m.alpha);
    ST(1)
    for (int l = 1; l < nhid; ++l) {
        const int ln = (l + 1 < nhid) ? l + 1 : l;
        const WSrc wn = make_wsrc(m.wp[ln] + (wave * NFT) * 256, lane);
        if constexpr (JM != 0)
            skew_layer<NFT, NKG, JM>(acc, wh, bA, wl, wn, make_bias(JM == 1 ? zero_bias : m.bp[l], bias_voff),
                                     make_bias(m.bp[ln], bias_voff), FTT * 64, bl, st_base, m.alpha ST_ARG,
                                     mask_ws ? mask_ws + l * (64 * NW) + wave * 64 + lane : nullptr,
                                     JM == 2 ? mask_wsB + l * (64 * NW) + wave * 64 + lane : nullptr);
        else
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
// A wave computes the NQW = 4 / NW quarters q0 .. q0 + NQW - 1 (q0 = wave * NQW), one accumulator chain each; NQL of them
// are not empty (all, except at width 64, whose sum is defined on the quarters of the 128-wide net: its last two
// contribute the +0 they contribute in the zero-padded 128-wide kernel).
template <int KQ, int NQL = 1>
__device__ __forceinline__ void small_load_w(const MlpArgs& m, int q0, int lane, f32x4 (&w)[2 * KQ * NQL]) {
    const f32x4* wp = (const f32x4*)m.wsmall + (size_t)(q0 * KQ) * 8 + (lane & 3);
#pragma unroll
    for (int i = 0; i < 2 * KQ * NQL; ++i) w[i] = wp[i * 4];
    __builtin_amdgcn_sched_barrier(0);
}
template <int KQ, int NQL = 1, int NQW = NQL, bool SUMQ = false>
__device__ __forceinline__ void last_layer_small(const MlpArgs& m, const f32x4 (&w)[2 * KQ * NQL], char* act, float* part,
                                                 int q0, int lane) {
    constexpr int MS = 64;
    f32x4 b[2 * KQ * NQL];
#pragma unroll
    for (int i = 0; i < 2 * KQ * NQL; ++i) b[i] = *(const f32x4*)(act + act_off<MS>(q0 * KQ + (i >> 1), lane, i & 1));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NQL];
#pragma unroll
    for (int q = 0; q < NQL; ++q) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 2 * KQ * q; i < 2 * KQ * (q + 1); ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i][e], b[i][e], acc[q], 0, 0, 0);
    }
    __syncthreads();   // every wave is done reading the activation image
    const int od = m.out_dim;
    if constexpr (SUMQ) {   // one wave holds every quarter: ((p0 + p1) + p2) + p3 here, one slot (the empty quarters' +0 left out)
        static_assert(NQW == 4, "SUMQ: one-wave workgroups");
        f32x4 sum = acc[0];
#pragma unroll
        for (int q = 1; q < NQL; ++q) sum += acc[q];
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < od) part[d * MS + lane] = sum[d];
        __syncthreads();
        return;
    }
#pragma unroll
    for (int q = 0; q < NQL; ++q)
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < od) part[((q0 + q) * od + d) * MS + lane] = acc[q][d];
    if constexpr (NQL < NQW) {
#pragma unroll
        for (int q = NQL; q < NQW; ++q)
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < od) part[((q0 + q) * od + d) * MS + lane] = 0.0f;
    }
    __syncthreads();
}

// Output layer of the one- and two-wave workgroups (NW = 1, 2), out_dim <= 32: this wave's NQL non-empty K-quarters one
// after the other, each its own accumulator chain on a 32-feature tile x 2 sample tiles, all kept in registers until
// every operand has been read (part aliases the activation image); its empty quarters are written as +0.
template <int KQ, int NQL, int NQW, bool SUMQ = false>
__device__ __forceinline__ void last_layer_quarters(const MlpArgs& m, char* act, float* part, int q0, int lane) {
    constexpr int MS = 64;
    const int h = lane >> 5, j = lane & 31;
    f32x16 acc[NQL][1][2];
#pragma unroll
    for (int q = 0; q < NQL; ++q) {
        const WSrc wl = make_wsrc(m.wp[m.n_layers - 1] + (size_t)((q0 + q) * KQ) * 256, lane);
        f32x4 a0[1], a1[1], a2[1];
        prefetch_a<1>(a0, a1, a2, wl, 64);
        gemm_tile<1, 2>(acc[q], a0, a1, a2, wl, 1, KQ, act + act_off<MS>((q0 + q) * KQ, j, h), nullptr);
    }
    __syncthreads();   // every operand has been read (one wave: just the LDS fence)
    const int od = m.out_dim;
    if constexpr (SUMQ) {   // one wave holds every quarter: ((p0 + p1) + p2) + p3 here, one slot (the empty quarters' +0 left out)
        static_assert(NQW == 4, "SUMQ: one-wave workgroups");
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16 sum = acc[0][0][st];
#pragma unroll
            for (int q = 1; q < NQL; ++q) sum += acc[q][0][st];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (d < od) part[d * MS + 32 * st + j] = sum[r];
            }
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (d < od) {
#pragma unroll
            for (int q = 0; q < NQW; ++q)
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    part[((q0 + q) * od + d) * MS + 32 * st + j] = (q < NQL) ? acc[q < NQL ? q : 0][0][st][r] : 0.0f;
        }
    }
    __syncthreads();
}

// Runs the MLP on the MS samples whose (padded) inputs sit in the x0 image.  On return
// part[(w*out_dim + d)*MS + s], w = 0..3, hold the four K-quarter partial sums of output d for sample s
// (bias not yet added).  Must be called by all 64 * NW threads of the workgroup.
// pre: see mlp_prefetch (call it for the same pass before mlp_core, as early as the caller can).
template <int HID, int NST, int JVP = 0>
constexpr bool mlp_is_skewed() { return NST == 2; }   // 64-sample passes: widths 64, 128, 256, 512 (the Jacobian passes too)

template <int HID, int NST, int JVP = 0, int NW = 4>
__device__ __forceinline__ void mlp_prefetch(const MlpArgs& m, MlpPre<HID, NW>& pre, int wave, int lane,
                                             const float* zero_bias = nullptr) {
    if constexpr (mlp_is_skewed<HID, NST, JVP>()) skew_prefetch<HID, NW>(m, pre, wave, lane, JVP == 1 ? zero_bias : nullptr);
}

// JVP (the Jacobian kernels, 64-sample passes only): 1 — sample tile 0 carries primal rows, tile 1 tangent columns of the
// same rows (zero_bias: the tangent tile's all-zero bias vector; mask_ws, optional: where the rows' slope masks go,
// [hidden layer][thread]); 2 — both tiles carry tangent columns, their slopes come from mask_ws (tile 0's rows) and
// mask_wsB (tile 1's rows) and m holds all-zero hidden biases.  See skew_layer.
// SUMQ (one-wave workgroups only; the fused kernel's choice): the output layer's four quarter sums are added in registers
// and only the sum goes to LDS — read it with mlp_out<MS, 1>.
template <int HID, int NST, int JVP = 0, int NW = 4, bool SUMQ = false>
__device__ __forceinline__ void mlp_core(const MlpArgs& m, MlpPre<HID, NW>& pre, char* x0, char* act, int wave,
                                         int lane ST_PARAM, unsigned long long* mask_ws = nullptr,
                                         const float* zero_bias = nullptr, const unsigned long long* mask_wsB = nullptr) {
    static_assert(JVP == 0 || NST == 2, "the Jacobian passes pair two 32-column tiles");
    static_assert(!SUMQ || NW == 1, "SUMQ: the wave must own all four quarters");
    static_assert(NW == 4 || (NST == 2 && ((NW == 1 && HID <= 128) || (NW == 2 && HID == 256))),
                  "one-wave workgroups: width <= 128; two-wave workgroups: width 256; both on 64-sample passes");
    constexpr int NFT = HID / (32 * NW);
    constexpr int FTT = HID / 32;
    constexpr int MS = 32 * NST;
    const int h = lane >> 5, j = lane & 31;
    const int nhid = m.n_layers - 1;
    // the output layer's K is split into four quarters whose partial sums are added ((p0+p1)+p2)+p3: KQ k-groups each
    // (at width 64 the quarters of the 128-wide net, as in the zero-padded 128-wide kernel and the CPU checker's kernel
    // order), NQW = 4 / NW of them per wave starting at q0, of which NQL are not empty.
    constexpr int KQ = (HID < 128 ? 128 : HID) / 32;
    constexpr int NQW = 4 / NW;
    constexpr int NQL = (NQW * KQ * 8 <= HID) ? NQW : HID / (KQ * 8);
    const int q0 = wave * NQW;
    const bool small_out = (NST == 2) && m.wsmall != nullptr;
    f32x4 wsm[2 * KQ * NQL];
    constexpr bool SKEWED = mlp_is_skewed<HID, NST, JVP>();
    if constexpr (SKEWED) {
        if constexpr (JVP != 0) hidden_skewed<HID, NW, JVP>(m, pre, x0, act, wave, lane ST_ARG, mask_ws, zero_bias, mask_wsB);
        else hidden_skewed<HID, NW>(m, pre, x0, act, wave, lane ST_ARG);
        ST(1)
        if (small_out) small_load_w<KQ, NQL>(m, q0, lane, wsm);
        __syncthreads();
        ST(1)
    } else {
        f32x16 acc[NFT][NST];
        f32x16 bv[NFT];
        load_bias<NFT>(bv, m.bp[0], wave, lane);
        gemm_tile_simple<NFT, NST>(acc, make_wsrc(m.wp[0] + (wave * NFT) * 256, lane), FTT, m.k0 >> 3,
                                        x0 + act_off<MS>(0, j, h), bv);
        // hidden layers hand sample tile 0 over already activated (final_stage_act0); layer 0 does it here
        constexpr bool ACT0 = (NST == 2);
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
            store_act<NFT, NST, ACT0>(acc, m.alpha, act, wave, lane);
            load_bias<NFT>(bv, m.bp[l + 1], wave, lane);
            ST(3)
            __syncthreads();
            ST(4)
            gemm_tile<NFT, NST, ACT0>(acc, a0, a1, a2, wnext, FTT, HID / 8, act + act_off<MS>(0, j, h), bv, m.alpha);
            ST(5)
        }
        // the last hidden layer's epilogue (kept out of the loop so the output layer's prefetched weights are not
        // live across the hidden layers' K loops)
        if (nhid > 1) __syncthreads();
        ST(2)
        store_act<NFT, NST, ACT0>(acc, m.alpha, act, wave, lane);
        if (small_out) small_load_w<KQ, NQL>(m, q0, lane, wsm);
        ST(3)
        __syncthreads();
        ST(4)
    }
    float* part = (float*)act;
    if (small_out) {
        last_layer_small<KQ, NQL, NQW, SUMQ>(m, wsm, act, part, q0, lane);
        ST(6)
        return;
    }
    if constexpr (NW != 4) {
        last_layer_quarters<KQ, NQL, NQW, SUMQ>(m, act, part, q0, lane);   // out_dim <= 32 (rt_set_mlp)
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

// NQ = 1 (mlp_core<..., SUMQ = true>): the one wave that computed all four quarters added them in registers, in this
// order, and left the sum in slot 0.
template <int MS, int NQ = 4>
__device__ __forceinline__ float mlp_out_nobias(const MlpArgs& m, const float* part, int d, int s) {
    const int od = m.out_dim;
    if constexpr (NQ == 1) return part[d * MS + s];
    const float p0 = part[(0 * od + d) * MS + s], p1 = part[(1 * od + d) * MS + s];
    const float p2 = part[(2 * od + d) * MS + s], p3 = part[(3 * od + d) * MS + s];
    return ((p0 + p1) + p2) + p3;
}
template <int MS, int NQ = 4>
__device__ __forceinline__ float mlp_out(const MlpArgs& m, const float* part, int d, int s) {
    return mlp_out_nobias<MS, NQ>(m, part, d, s) + m.bp[m.n_layers - 1][d];
}

