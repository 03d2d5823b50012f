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
//     feature slice straight global(L2) -> VGPR with 16-byte coalesced buffer loads, software-pipelined three
//     k-groups ahead; no LDS staging of weights;
//   * the proposal sampler (Philox4x32-10 + Box-Muller), the Gaussian / Student-t scoring (fp64) and a per-pass
//     log-sum-exp pair are fused around the MLP, so nothing per-sample ever reaches HBM; the pairs of a query point
//     are merged in a fixed order by a small second kernel, which makes results independent of the work partition.
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
#include <mutex>
#include <type_traits>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl.so.1 (573 MB) is dlopen'ed by rt_comm_init, never linked

#include "../../include/rt_is.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#include "rt_mlp.hpp"
#include "rt_sampler.hpp"
#include "rt_kernels.hpp"

// ----------------------------------------------------------------------------------------------
// Scratch of the no-allocation entry point (rt_is_estimate_dev_noalloc): the fused kernel's per-pass (max, sum) pairs go
// here instead of into a hipMalloc'ed buffer.  One copy per GPU, held until the process ends.  In host code the name
// `scratch` is only a handle for hipGetSymbolAddress (rt_ctx_create, once per device); everything else uses the device
// address kept in g_scratch_ptr / rt_ctx::scratch.
// ----------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];      // device address of `scratch` on HIP device i (nullptr: not looked up yet)
std::mutex g_scratch_mu;
// Who used device i's array last, and the event that context records behind its work on it (its own, made when the context
// was: recording allocates nothing).  A second context asking for the array while that event has not completed is refused
// (RT_EBUSY) instead of silently racing for the pairs.
struct rt_ctx* g_scratch_owner[MAX_DEVICES];
hipEvent_t g_scratch_event[MAX_DEVICES];
// XCD shares (balance_update_kernel) a destroyed context of this process left for HIP device i: the relative speeds of a
// chip's XCDs are a property of the chip, so the next context on it starts from them instead of from equal shares
double g_bal_seed[MAX_DEVICES][9];
bool g_bal_seeded[MAX_DEVICES];

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
    int nw = 4;               // waves per workgroup (rt_mlp.hpp): 1 for narrow nets (width <= 128, small in / out dims)
    std::vector<int> dims;
    std::vector<void*> dev;   // device allocations to free
    MlpArgs args;
    MlpArgs args0;            // args with all-zero hidden-layer biases: the tangent-only passes (mlp_jvp_rest_kernel)
    double flops_per_row = 0.0;
};

}  // namespace

struct rt_ctx {
    int device = 0;
    int cu_count = 0;
    hipStream_t stream = nullptr;
    HostMlp mlp[2];
    DevBuf b_y, b_xhat, b_xs, b_out, b_partial, b_in, b_fout, b_csend, b_crecv, b_sum, b_mask;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<std::pair<const void*, int>> lds_allowed;   // kernels whose dynamic-LDS limit has been raised, and to what
    size_t events_used = 0;
    double ms_harvested = 0.0;              // durations already folded in by harvest_events()
    uint64_t launches_harvested = 0;
    void* scratch = nullptr;                // this device's copy of the `scratch` array (see above)
    hipEvent_t scratch_done = nullptr;      // recorded behind every rt_is_estimate_dev_noalloc call (see g_scratch_owner)
    double* d_bal = nullptr;                // XCD balance: 9 cumulative shares (balance_update_kernel) ...
    unsigned long long* d_wg_times = nullptr;   // ... and the workgroups' start / end times of the last launch (2 x 1024)
    bool balance = true;                    // RT_IS_BALANCE=0 (read by rt_ctx_create) turns the balancing off
    uint64_t bal_updates = 0;               // share updates enqueued so far (a context seeded from an earlier one starts at 1)
    bool first_block = true;                // RT_IS_FIRST_BLOCK=0 (read by rt_ctx_create): no measuring first block (is_enqueue)
    ncclComm_t comm = nullptr;              // rt_comm_init
    int comm_rank = 0, comm_nranks = 1;
    int64_t max_launch_evals = 1ll << 28;   // sample-evals per fused-kernel launch (RT_IS_MAX_LAUNCH_EVALS overrides)
    int jvp_masks = 1;                      // RT_JVP_MASKS (read by rt_ctx_create): 0 the Jacobian recomputes the primal per direction,
                                            // 1 (default) whichever kernel is faster, 2 always the mask kernel (tests)
    int64_t cf_block_points = 1 << 20;      // query points per block of the Laplace estimator (RT_CF_BLOCK_POINTS overrides)
    double flops_acc = 0.0;
    uint64_t evals_acc = 0;
    std::string err;
};

namespace {

thread_local std::string g_create_err;   // rt_ctx_create has no context to hang its error text on

constexpr size_t MAX_EVENT_PAIRS = 512;   // bound on live HIP events per context (see harvest_events)

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

// Folds the recorded launch durations into ctx->ms_harvested and frees the event pairs for reuse.  Waits for the last
// recorded launch: called by the counter readers, and by the enqueue path once MAX_EVENT_PAIRS launches are outstanding
// so that a caller who never reads the counters (a serving loop on rt_is_estimate_dev) holds a bounded number of events.
int harvest_events(rt_ctx* ctx) {
    if (ctx->events_used == 0) return RT_OK;
    HIPCHK(ctx, hipEventSynchronize(ctx->events[ctx->events_used - 1].second));
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float t = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&t, ctx->events[i].first, ctx->events[i].second));
        ctx->ms_harvested += t;
    }
    ctx->launches_harvested += ctx->events_used;
    ctx->events_used = 0;
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

// Raises a kernel's dynamic-LDS limit to `lds` bytes, once per (context, kernel): the attribute is per device and
// sticky, and a context stays on one device, so only growth needs another hipFuncSetAttribute.
hipError_t allow_lds(rt_ctx* ctx, const void* fn, int lds) {
    for (auto& kv : ctx->lds_allowed)
        if (kv.first == fn) {
            if (kv.second >= lds) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) kv.second = lds;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) ctx->lds_allowed.emplace_back(fn, lds);
    return e;
}

template <int HID, int NST, bool GIVEN, bool SHARE, int NW>
int launch_is_variant(rt_ctx* ctx, const IsArgs& a, int grid, int lds) {
    HIPCHK(ctx, allow_lds(ctx, (const void*)is_fused_kernel<HID, NST, GIVEN, SHARE, NW>, lds));
    hipLaunchKernelGGL((is_fused_kernel<HID, NST, GIVEN, SHARE, NW>), dim3(grid), dim3(64 * NW), lds, ctx->stream, a);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

template <int HID, int NST, int NW = 4>
int launch_is(rt_ctx* ctx, const IsArgs& a, bool given, int grid, int lds) {
    if (given) return launch_is_variant<HID, NST, true, false, NW>(ctx, a, grid, lds);
    // more than one Philox call per sample (4 normals each: df of them for chi2, then the dx coordinates; the gamma path
    // of a non-integer df draws its chi2 apart): the threads of a sample slot share the calls.  (A one-wave workgroup
    // has one thread per sample: nothing to share.)
    const bool share = NW > 1 && a.sa.ndf + a.g.in_dim > 4;
    if constexpr (NW > 1) {
        if (share) return launch_is_variant<HID, NST, false, true, NW>(ctx, a, grid, lds);
    }
    return launch_is_variant<HID, NST, false, false, NW>(ctx, a, grid, lds);
}

template <int HID, int NST, int NW = 4>
int launch_fwd(rt_ctx* ctx, const MlpArgs& m, const float* in, int64_t N, float* out, int grid, int lds) {
    HIPCHK(ctx, allow_lds(ctx, (const void*)mlp_fwd_kernel<HID, NST, NW>, lds));
    hipLaunchKernelGGL((mlp_fwd_kernel<HID, NST, NW>), dim3(grid), dim3(64 * NW), lds, ctx->stream, m, in, N, out);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

template <int HID, int NW = 4>
int launch_jvp(rt_ctx* ctx, const HostMlp& M, const float* in, int64_t N, float* yout, float* J, int grid, int lds) {
    HIPCHK(ctx, allow_lds(ctx, (const void*)mlp_jvp_kernel<HID, NW>, lds));
    hipLaunchKernelGGL((mlp_jvp_kernel<HID, NW>), dim3(grid), dim3(64 * NW), lds, ctx->stream, M.args, M.args0.bp[0], in, N, yout, J);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

int make_sampler(rt_ctx* ctx, double scale, double df, uint64_t seed, SamplerArgs& sa) {
    if (!(scale > 0.0) || !(df > 0.0) || !isfinite(scale) || !isfinite(df))
        return fail(ctx, RT_EINVAL, "scale and df must be positive and finite");
    sa.seed_lo = (uint32_t)seed;
    sa.seed_hi = (uint32_t)(seed >> 32);
    sa.df = (float)df;
    sa.inv_df = (float)(1.0 / df);
    sa.scale = (float)scale;
    sa.ndf = (df == floor(df) && df <= 32.0) ? (int)df : 0;
    return RT_OK;
}

// How one launch over N query points x S samples is laid out as passes (see the comment above is_fused_kernel).
struct IsPlan {
    int64_t ppp, p_main, p_total, E;
    int rem, q;
};
IsPlan is_plan(int64_t N, int64_t S, int nst) {
    const int MS = 32 * nst;
    const int64_t full = S / MS;
    const int r = (int)(S - full * MS);
    const bool packed = r > 0 && 2 * r <= MS;      // at least two query points fit into one leftover pass
    IsPlan pl;
    pl.ppp = packed ? full : (S + MS - 1) / MS;
    pl.rem = packed ? r : 0;
    pl.q = packed ? MS / r : 0;
    pl.p_main = N * pl.ppp;
    pl.p_total = pl.p_main + (packed ? (N + pl.q - 1) / pl.q : 0);
    pl.E = pl.ppp + (packed ? 1 : 0);
    return pl;
}
size_t is_partial_bytes(int64_t N, int64_t S, int nst, int n_sd) {
    return (size_t)N * (size_t)is_plan(N, S, nst).E * n_sd * 2 * sizeof(double);
}

// Where a launch's per-pass pairs go, and whether the launch is bracketed by HIP events for rt_get_counters: the
// context's own buffer (timed), or the device's `scratch` array for rt_is_estimate_dev_noalloc (untimed — creating an
// event allocates).
struct PairBuf {
    double* p;
    size_t cap;
    bool timed;
};

// Enqueues one launch of the fused kernel for N query points, computing the pairs [e_lo, e_hi) of every point into
// pb.p (laid out N x E x n_sd x 2 whatever the range; e_hi < 0 means all E), and — when out is not NULL — the
// finalize kernel whose results go to out[k*out_stride + n] (only meaningful for the full range).
// All pointers are device pointers.
int is_enqueue_block(rt_ctx* ctx, const PairBuf& pb, const double* y, const float* xhat, const double* xs, int64_t N,
                     int64_t y_index_base, int64_t S, int n_sd, const double* sd_ys, double scale, double df,
                     uint64_t seed, int log_flag, double* out, int64_t out_stride,
                     int64_t e_lo = 0, int64_t e_hi = -1) {
    HostMlp& G = ctx->mlp[RT_MLP_G];
    IsArgs a;
    memset(&a, 0, sizeof(a));
    a.g = G.args;
    int rc = make_sampler(ctx, scale, df, seed, a.sa);
    if (rc) return rc;
    const int dx = G.args.in_dim, dy = G.args.out_dim;
    const int nst = G.nst;
    const int MS = 32 * nst;
    const int grid_max = ctx->cu_count * (4 / G.nw);   // one- / two-wave workgroups: four / two per CU, a wave per SIMD
    const IsPlan pl = is_plan(N, S, nst);
    if (e_hi < 0) e_hi = pl.E;
    // main entries of the range, and whether the leftover entry (the last one, E - 1, in packed mode) is in it
    const int64_t m_hi = e_hi < pl.ppp ? e_hi : pl.ppp;
    const int64_t cnt = m_hi > e_lo ? m_hi - e_lo : 0;
    const bool do_left = pl.rem > 0 && e_lo < pl.E && e_hi == pl.E;
    a.ppp = cnt; a.j_lo = cnt ? e_lo : 0; a.left_s0 = pl.ppp * MS;
    a.p_main = N * cnt; a.p_total = a.p_main + (do_left ? pl.p_total - pl.p_main : 0);
    a.E = pl.E; a.rem = do_left ? pl.rem : 0; a.q = do_left ? pl.q : 0;
    if (a.p_total == 0) return RT_OK;           // an empty range: nothing to run
    a.N = N; a.S = S; a.y_index_base = y_index_base;
    a.y = y; a.xhat = xhat; a.xs = xs;
    a.n_sd = n_sd;
    if (is_partial_bytes(N, S, nst, n_sd) > pb.cap)
        return fail(ctx, RT_EINVAL, "internal: partial-result buffer smaller than the block needs");
    a.partial = pb.p;
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
    // (set below, once the grid is known: a.bal, a.wg_times)

    const Carve cv = carve_for(G.hid, MS, G.args.k0, dx, dy, 64 * G.nw);
    const int grid = (int)((a.p_total < grid_max) ? a.p_total : grid_max);
    // XCD balance: only launches that fill the chip (grid a multiple of 8, <= 1024 workgroups) use the shares, and only
    // long ones (>= 256 passes per workgroup: a pass of granularity is then < 0.4 % of the launch) update them
    const bool use_bal = ctx->balance && ctx->d_bal && grid == grid_max && grid % 8 == 0 && grid <= 1024;
    const bool upd_bal = use_bal && a.p_total / grid >= 256;
    a.bal = use_bal ? ctx->d_bal : nullptr;
    a.wg_times = upd_bal ? ctx->d_wg_times : nullptr;
    if (pb.timed) {
        if (ctx->events_used == MAX_EVENT_PAIRS) {
            rc = harvest_events(ctx);
            if (rc) return rc;
        }
        if (ctx->events_used == ctx->events.size()) {
            hipEvent_t e0, e1;
            HIPCHK(ctx, hipEventCreate(&e0));
            HIPCHK(ctx, hipEventCreate(&e1));
            ctx->events.emplace_back(e0, e1);
        }
        HIPCHK(ctx, hipEventRecord(ctx->events[ctx->events_used].first, ctx->stream));
    }
    const bool given = xs != nullptr;
    // one-wave (two-wave) workgroups ask for more than a fifth (a third) of the LDS, so that at most four (two) share a
    // CU — a wave per SIMD — however few registers and however little LDS they need
    const int lds1 = cv.total > 33 * 1024 ? cv.total : 33 * 1024;
    const int lds2 = cv.total > 54 * 1024 ? cv.total : 54 * 1024;
    switch (G.hid * 10 + nst + (G.nw != 4 ? 100000 * G.nw : 0)) {
        case 100642: rc = launch_is<64, 2, 1>(ctx, a, given, grid, lds1); break;
        case 101282: rc = launch_is<128, 2, 1>(ctx, a, given, grid, lds1); break;
        case 202562: rc = launch_is<256, 2, 2>(ctx, a, given, grid, lds2); break;
        case 1282: rc = launch_is<128, 2>(ctx, a, given, grid, cv.total); break;
        case 2562: rc = launch_is<256, 2>(ctx, a, given, grid, cv.total); break;
        case 5122: rc = launch_is<512, 2>(ctx, a, given, grid, cv.total); break;
        case 10241: rc = launch_is<1024, 1>(ctx, a, given, grid, cv.total); break;
        default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
    }
    if (rc) return rc;
    if (pb.timed) {   // sample-evals of this launch (the whole of S unless the call covers a range of a point's passes)
        HIPCHK(ctx, hipEventRecord(ctx->events[ctx->events_used].second, ctx->stream));
        ctx->events_used++;
        int64_t per_point = cnt * MS + (do_left ? pl.rem : 0);
        if (pl.rem == 0 && cnt && e_lo + cnt == pl.ppp) per_point -= pl.ppp * MS - S;   // the part-full last main pass
        ctx->flops_acc += (double)N * (double)per_point * G.flops_per_row;
        ctx->evals_acc += (uint64_t)N * (uint64_t)per_point;
    }
    if (upd_bal) {
        hipLaunchKernelGGL(balance_update_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_bal,
                           (const unsigned long long*)ctx->d_wg_times, grid);
        HIPCHK(ctx, hipGetLastError());
        ctx->bal_updates++;
    }
    if (!out) return RT_OK;
    hipLaunchKernelGGL(is_finalize_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const double*)a.partial, N, a.E, n_sd, log((double)S), log_flag, out, out_stride);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

// Validates, then enqueues the estimate as one launch per block of query points: a launch is bounded to
// ctx->max_launch_evals sample-evals (about 8 s of one MI355X by default), so that a 10^10-eval call (BASELINE config 3's
// share of one GPU) is a sequence of bounded kernels rather than one that runs for minutes.  Query points are
// independent and the sampler is keyed by the global index, so the split does not change any draw.
// fixed == nullptr: the pairs go to the context's own buffer, grown (hipMalloc) to the largest block.
// fixed != nullptr (rt_is_estimate_dev_noalloc): they go to that buffer, the call is refused if the pairs of ALL its
// query points exceed fixed->cap, and nothing on this path allocates (no hipMalloc / hipFree / hipEventCreate).
int is_enqueue(rt_ctx* ctx, const double* y, const float* xhat, const double* xs, int64_t N,
               int64_t y_index_base, int64_t S, int n_sd, const double* sd_ys, double scale, double df,
               uint64_t seed, int log_flag, double* out, const PairBuf* fixed = nullptr) {
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set");
    if (N < 0 || S <= 0 || S > 0xFFFFFFFFll) return fail(ctx, RT_EINVAL, "need N >= 0 and 0 < S < 2^32");
    if (n_sd < 1 || n_sd > RT_MAX_SD || !sd_ys) return fail(ctx, RT_EINVAL, "n_sd must be in [1, RT_MAX_SD]");
    for (int k = 0; k < n_sd; ++k)
        if (!(sd_ys[k] > 0.0) || !isfinite(sd_ys[k])) return fail(ctx, RT_EINVAL, "sd_y must be positive and finite");
    if (N == 0) return RT_OK;
    const int dx = G.args.in_dim, dy = G.args.out_dim;
    if (fixed) {
        const size_t one = is_partial_bytes(1, S, G.nst, n_sd);
        if ((size_t)N > fixed->cap / one) {
            char msg[256];
            snprintf(msg, sizeof msg, "rt_is_estimate_dev_noalloc: N = %lld query points x %lld (max, sum) pairs x 16 B exceed "
                     "the %u-byte scratch array (RT_NOALLOC_SCRATCH_BYTES): at this sample_size at most %lld query points "
                     "per call", (long long)N, (long long)(one / 16), (unsigned)RT_NOALLOC_SCRATCH_BYTES,
                     (long long)(fixed->cap / one));
            return fail(ctx, RT_ETOOBIG, msg);
        }
    }
    int64_t per = ctx->max_launch_evals / S;
    {   // ... and to 1 GiB of per-pass partial results (16 B per pass and sd value; only matters for tiny S)
        const int64_t per_mem = (int64_t)((1ull << 30) / is_partial_bytes(1, S, G.nst, n_sd));
        if (per > per_mem) per = per_mem;
    }
    if (per < 1) per = 1;
    const int64_t nblocks = (N + per - 1) / per;
    per = (N + nblocks - 1) / nblocks;           // balanced blocks
    // the partial-result buffer is sized ONCE for the largest block, before anything is enqueued: growing it between
    // two launches of one call would free memory the previous launch may still be writing
    if (!fixed) {
        size_t need = 0;
        for (int64_t n0 = 0; n0 < N; n0 += per) {
            const int64_t nb = (N - n0 < per) ? N - n0 : per;
            const size_t b = is_partial_bytes(nb, S, G.nst, n_sd);
            if (b > need) need = b;
        }
        const int rc = ensure(ctx, ctx->b_partial, need);
        if (rc) return rc;
    }
    // (blocks of one call reuse the buffer from its start: they run in stream order)
    const PairBuf pb = fixed ? *fixed : PairBuf{(double*)ctx->b_partial.p, ctx->b_partial.cap, true};
    int64_t n_first = 0;
    if (ctx->balance && ctx->first_block && ctx->bal_updates == 0 && !xs) {
        // The XCD shares are still the equal ones (first long call of the process on this GPU): a call that is at least
        // four times a share update's minimum runs that minimum — 320 passes per workgroup, about 0.15 s — as a block of
        // its own first, so that the rest of the call already runs on measured shares (0.7 % of its time) instead of
        // only the next call.  Blocks are independent: no result changes.  (RT_IS_FIRST_BLOCK=0 keeps such a call in one
        // launch: bench.py sets it so that every step stays ONE launch of the kernel its rocprofv3 summary averages.)
        const IsPlan p1 = is_plan(1, S, G.nst);
        const int64_t grid_max = (int64_t)ctx->cu_count * (4 / G.nw);
        const int64_t n_cal = (320 * grid_max + p1.ppp - 1) / (p1.ppp > 0 ? p1.ppp : 1);
        const bool can_balance = ctx->d_bal && grid_max % 8 == 0 && grid_max <= 1024;   // is_enqueue_block's conditions
        if (can_balance && p1.ppp > 0 && n_cal <= per && 4 * n_cal <= N) {
            const int rc = is_enqueue_block(ctx, pb, y, xhat, nullptr, n_cal, y_index_base, S, n_sd, sd_ys, scale, df, seed,
                                            log_flag, out, N);
            if (rc) return rc;
            n_first = n_cal;
        }
    }
    for (int64_t n0 = n_first; n0 < N; n0 += per) {
        const int64_t nb = (N - n0 < per) ? N - n0 : per;
        const int rc = is_enqueue_block(ctx, pb, y + n0 * dy, xhat + n0 * dx, xs ? xs + (size_t)n0 * S * dx : nullptr, nb,
                                        y_index_base + n0, S, n_sd, sd_ys, scale, df, seed, log_flag, out + n0, N);
        if (rc) return rc;
    }
    return RT_OK;
}

int fwd_enqueue(rt_ctx* ctx, int which, const float* in, int64_t N, float* out) {
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    HostMlp& M = ctx->mlp[which];
    if (!M.set) return fail(ctx, RT_ENOTSET, "MLP has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    const int ms = 32 * M.nst;
    const Carve cv = carve_for(M.hid, ms, M.args.k0, M.args.in_dim, M.args.out_dim, 64 * M.nw);
    const int64_t npass = (N + ms - 1) / ms;
    const int64_t gmax = (int64_t)ctx->cu_count * (M.nw == 1 ? 4 : 2);
    const int grid = (int)(npass < gmax ? npass : gmax);
    if (M.nw == 1) {
        if (M.hid == 64) return launch_fwd<64, 2, 1>(ctx, M.args, in, N, out, grid, cv.total);
        return launch_fwd<128, 2, 1>(ctx, M.args, in, N, out, grid, cv.total);
    }
    if (M.nw == 2) return launch_fwd<256, 2, 2>(ctx, M.args, in, N, out, grid, cv.total);
    switch (M.hid) {
        case 128: return launch_fwd<128, 2>(ctx, M.args, in, N, out, grid, cv.total);
        case 256: return launch_fwd<256, 2>(ctx, M.args, in, N, out, grid, cv.total);
        case 512: return launch_fwd<512, 2>(ctx, M.args, in, N, out, grid, cv.total);
        case 1024: return launch_fwd<1024, 1>(ctx, M.args, in, N, out, grid, cv.total);
        default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
    }
}

template <int HID, int NW = 4>
int launch_jvp2(rt_ctx* ctx, const HostMlp& M, const float* in, int64_t N, float* yout, float* J, int64_t gmax, int lds) {
    const int64_t nblk = (N + 31) / 32, nrest = (nblk * (M.args.in_dim - 1) + 1) / 2;
    unsigned long long* masks = (unsigned long long*)ctx->b_mask.p;
    HIPCHK(ctx, allow_lds(ctx, (const void*)mlp_jvp_first_kernel<HID, NW>, lds));
    hipLaunchKernelGGL((mlp_jvp_first_kernel<HID, NW>), dim3((unsigned)(nblk < gmax ? nblk : gmax)), dim3(64 * NW), lds,
                       ctx->stream, M.args, M.args0.bp[0], in, N, yout, J, masks);
    HIPCHK(ctx, hipGetLastError());
    if (nrest == 0) return RT_OK;
    HIPCHK(ctx, allow_lds(ctx, (const void*)mlp_jvp_rest_kernel<HID, NW>, lds));
    hipLaunchKernelGGL((mlp_jvp_rest_kernel<HID, NW>), dim3((unsigned)(nrest < gmax ? nrest : gmax)), dim3(64 * NW), lds,
                       ctx->stream, M.args0, N, J, (const unsigned long long*)masks);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

int jvp_enqueue(rt_ctx* ctx, int which, const float* in, int64_t N, float* yout, float* J) {
    if (which != RT_MLP_G && which != RT_MLP_H) return fail(ctx, RT_EINVAL, "which must be RT_MLP_G or RT_MLP_H");
    HostMlp& M = ctx->mlp[which];
    if (!M.set) return fail(ctx, RT_ENOTSET, "MLP has not been set");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N == 0) return RT_OK;
    if (M.nst != 2) return fail(ctx, RT_EUNSUPPORTED, "Jacobian needs hidden width <= 512 (64-column passes)");
    const Carve cv = carve_for(M.hid, 64, M.args.k0, M.args.in_dim, M.args.out_dim, 64 * M.nw);
    const int64_t nblk = (N + 31) / 32;
    const int64_t npass = nblk * M.args.in_dim;
    const int64_t gmax = (int64_t)ctx->cu_count * (M.nw == 1 ? 4 : 2);
    const int grid = (int)(npass < gmax ? npass : gmax);
    {   // The primal once per block of rows (mlp_jvp_first_kernel + mlp_jvp_rest_kernel: (1 + dx) / 2 passes per block)
        // or once per direction (mlp_jvp_kernel: dx passes per block): whichever leaves the busiest workgroup fewer
        // passes.  RT_JVP_MASKS=0 (read by rt_ctx_create) keeps the per-direction kernel, =2 forces the other.
        const int64_t nrest = (nblk * (M.args.in_dim - 1) + 1) / 2;
        const int64_t g1 = nblk < gmax ? nblk : gmax, g2 = nrest < gmax ? nrest : gmax;
        const int64_t busiest2 = (nblk + g1 - 1) / g1 + (g2 ? (nrest + g2 - 1) / g2 : 0), busiest1 = (npass + grid - 1) / grid;
        const size_t mask_bytes = (size_t)nblk * (M.n_layers - 1) * 64 * M.nw * sizeof(unsigned long long);
        if ((ctx->jvp_masks == 1 && M.args.in_dim >= 2 && busiest2 < busiest1 && mask_bytes <= (1ull << 30)) ||
            (ctx->jvp_masks == 2 && M.args.in_dim >= 2)) {
            const int rc = ensure(ctx, ctx->b_mask, mask_bytes);
            if (rc) return rc;
            if (M.nw == 1) {
                if (M.hid == 64) return launch_jvp2<64, 1>(ctx, M, in, N, yout, J, gmax, cv.total);
                return launch_jvp2<128, 1>(ctx, M, in, N, yout, J, gmax, cv.total);
            }
            if (M.nw == 2) return launch_jvp2<256, 2>(ctx, M, in, N, yout, J, gmax, cv.total);
            switch (M.hid) {
                case 128: return launch_jvp2<128>(ctx, M, in, N, yout, J, gmax, cv.total);
                case 256: return launch_jvp2<256>(ctx, M, in, N, yout, J, gmax, cv.total);
                case 512: return launch_jvp2<512>(ctx, M, in, N, yout, J, gmax, cv.total);
                default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
            }
        }
    }
    if (M.nw == 1) {
        if (M.hid == 64) return launch_jvp<64, 1>(ctx, M, in, N, yout, J, grid, cv.total);
        return launch_jvp<128, 1>(ctx, M, in, N, yout, J, grid, cv.total);
    }
    if (M.nw == 2) return launch_jvp<256, 2>(ctx, M, in, N, yout, J, grid, cv.total);
    switch (M.hid) {
        case 128: return launch_jvp<128>(ctx, M, in, N, yout, J, grid, cv.total);
        case 256: return launch_jvp<256>(ctx, M, in, N, yout, J, grid, cv.total);
        case 512: return launch_jvp<512>(ctx, M, in, N, yout, J, grid, cv.total);
        default: return fail(ctx, RT_EUNSUPPORTED, "hidden width not supported");
    }
}

// The Laplace estimator for N query points whose float64 coordinates sit at y (device): per block of points, in stream
// order, y -> float32 feed -> H forward (xhat) -> G forward + Jacobian at xhat (mlp_jvp_kernel) -> cf_finish_kernel.
// The Jacobians never leave the device: a block's (<= 1 GiB of float32, N_blk x dy x dx) is written by one kernel and
// read by the next, which at the reference shapes is < 0.5 % of the Jacobian kernel's own time (YearPredictionMSD 20/90:
// 7.2 KB per point against 22 generator rows x 4.8 MFLOP).  Buffers are sized once, before anything is enqueued.
int cf_enqueue(rt_ctx* ctx, const double* y, int64_t N, double sd_y, int log_flag, double* out) {
    HostMlp& G = ctx->mlp[RT_MLP_G];
    HostMlp& H = ctx->mlp[RT_MLP_H];
    if (!G.set || !H.set) return fail(ctx, RT_ENOTSET, "the Laplace estimator needs both G (RT_MLP_G) and H (RT_MLP_H)");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (!(sd_y > 0.0) || !isfinite(sd_y)) return fail(ctx, RT_EINVAL, "sd_y must be positive and finite");
    const int dx = G.args.in_dim, dy = G.args.out_dim;
    if (H.args.in_dim != dy || H.args.out_dim != dx)
        return fail(ctx, RT_EINVAL, "H must map G's output dim to G's input dim");
    if (G.nst != 2) return fail(ctx, RT_EUNSUPPORTED, "Jacobian needs hidden width <= 512 (64-column passes)");
    int lds = (dx * dx + dy + dx) * (int)sizeof(double);
    if (lds > LDS_MAX) return fail(ctx, RT_EUNSUPPORTED, "Laplace estimator: dx x dx float64 matrix does not fit the LDS");
    // cf_finish_kernel keeps the point's Jacobian in LDS as well where that leaves several one-wave workgroups to a CU
    const bool stage = lds + dx * dy * (int)sizeof(float) <= 40 * 1024;
    if (stage) lds += dx * dy * (int)sizeof(float);
    const void* finish = stage ? (const void*)cf_finish_kernel<true> : (const void*)cf_finish_kernel<false>;
    if (N == 0) return RT_OK;
    int64_t per = (int64_t)((1ull << 30) / ((size_t)dx * dy * sizeof(float)));
    if (per > ctx->cf_block_points) per = ctx->cf_block_points;
    {   // ... and <= 1 GiB of the Jacobian kernels' slope masks (8 B per thread and hidden layer for every 32 rows)
        const int64_t per_mask = (int64_t)((1ull << 30) / ((size_t)(G.n_layers - 1) * 64 * G.nw * sizeof(unsigned long long))) * 32;
        if (per > per_mask) per = per_mask;
    }
    per &= ~(int64_t)31;                       // whole 32-row blocks of the Jacobian kernel
    if (per < 32) per = 32;
    if (per > N) per = N;
    int rc = ensure(ctx, ctx->b_in, (size_t)per * dy * sizeof(float)); if (rc) return rc;
    rc = ensure(ctx, ctx->b_xhat, (size_t)per * dx * sizeof(float)); if (rc) return rc;
    rc = ensure(ctx, ctx->b_fout, (size_t)per * dy * (1 + (size_t)dx) * sizeof(float)); if (rc) return rc;
    HIPCHK(ctx, allow_lds(ctx, finish, lds));
    const double PI = 3.14159265358979323846;
    // the constants of :382 / :384 as the reference's Python evaluates them
    const double c0 = -(double)dy * log(sqrt(2.0 * PI) * sd_y);
    const double lin_scale = 1.0 / pow(sqrt(2.0 * PI) * sd_y, (double)dy) * pow(sd_y, (double)dy);
    float* y32 = (float*)ctx->b_in.p;
    float* xhat = (float*)ctx->b_xhat.p;
    float* gy = (float*)ctx->b_fout.p;
    float* J = gy + (size_t)per * dy;
    for (int64_t n0 = 0; n0 < N; n0 += per) {
        const int64_t nb = (N - n0 < per) ? N - n0 : per;
        const int64_t cnt = nb * dy;
        hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                           y + n0 * dy, cnt, y32);
        HIPCHK(ctx, hipGetLastError());
        rc = fwd_enqueue(ctx, RT_MLP_H, y32, nb, xhat); if (rc) return rc;
        rc = jvp_enqueue(ctx, RT_MLP_G, xhat, nb, gy, J); if (rc) return rc;
        const int64_t gmax = (int64_t)ctx->cu_count * 16;
        const dim3 fgrid((unsigned)(nb < gmax ? nb : gmax));
        if (stage)
            hipLaunchKernelGGL(cf_finish_kernel<true>, fgrid, dim3(64), lds, ctx->stream, y + n0 * dy, (const float*)xhat,
                               (const float*)gy, (const float*)J, nb, dx, dy, sd_y * sd_y, c0, lin_scale, log_flag, out + n0);
        else
            hipLaunchKernelGGL(cf_finish_kernel<false>, fgrid, dim3(64), lds, ctx->stream, y + n0 * dy, (const float*)xhat,
                               (const float*)gy, (const float*)J, nb, dx, dy, sd_y * sd_y, c0, lin_scale, log_flag, out + n0);
        HIPCHK(ctx, hipGetLastError());
    }
    return RT_OK;
}

// (mean, standard error) of the N float64 values at v (device) into out2 (device, 2 doubles), in numpy's summation
// order (np_chunk_sum_kernel).  b_sum holds the per-buffer sums followed by the two results when out2 is NULL.
int mean_stderr_enqueue(rt_ctx* ctx, const double* v, int64_t N, double* out2) {
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    const int64_t nchunks = (N + NP_BUF - 1) / NP_BUF;
    const int rc = ensure(ctx, ctx->b_sum, (size_t)(nchunks + 2) * sizeof(double));
    if (rc) return rc;
    double* cs = (double*)ctx->b_sum.p;
    if (!out2) out2 = cs + nchunks;
    for (int stage = 0; stage < 2; ++stage) {
        if (nchunks > 0) {
            if (stage == 0)
                hipLaunchKernelGGL(np_chunk_sum_kernel<false>, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, v, N,
                                   (const double*)nullptr, cs);
            else
                hipLaunchKernelGGL(np_chunk_sum_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, v, N,
                                   (const double*)out2, cs);
            HIPCHK(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(np_reduce_kernel, dim3(1), dim3(1), 0, ctx->stream, (const double*)cs, nchunks, N, stage, out2);
        HIPCHK(ctx, hipGetLastError());
    }
    return RT_OK;
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
    if (const char* v = getenv("RT_IS_MAX_LAUNCH_EVALS")) {
        const long long n = atoll(v);
        if (n > 0) ctx->max_launch_evals = n;
    }
    if (const char* v = getenv("RT_JVP_MASKS")) ctx->jvp_masks = (v[0] == '0') ? 0 : (v[0] == '2') ? 2 : 1;
    if (const char* v = getenv("RT_CF_BLOCK_POINTS")) {
        const long long n = atoll(v);
        if (n > 0) ctx->cf_block_points = n;
    }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, RT_EHIP, hipGetErrorString(e)); }
    ctx->lds_allowed.reserve(64);    // so that no launch path grows it (rt_is_estimate_dev_noalloc allocates nothing)
    {   // XCD balance state (see balance_update_kernel): equal shares to start with
        if (const char* v = getenv("RT_IS_BALANCE")) ctx->balance = !(v[0] == '0');
        if (const char* v = getenv("RT_IS_FIRST_BLOCK")) ctx->first_block = !(v[0] == '0');
        void* p = nullptr;
        e = hipMalloc(&p, 16 * sizeof(double) + 2 * 1024 * sizeof(unsigned long long));
        if (e == hipSuccess) {
            double init[16] = {0};
            for (int i = 0; i <= 8; ++i) init[i] = (double)i / 8.0;
            if (ctx->balance && device < MAX_DEVICES) {
                std::lock_guard<std::mutex> lk(g_scratch_mu);
                if (g_bal_seeded[device]) { memcpy(init, g_bal_seed[device], sizeof g_bal_seed[device]); ctx->bal_updates = 1; }
            }
            e = hipMemcpy(p, init, sizeof init, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            if (p) (void)hipFree(p);
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail(nullptr, RT_EHIP, std::string("XCD balance state: ") + hipGetErrorString(e));
        }
        ctx->d_bal = (double*)p;
        ctx->d_wg_times = (unsigned long long*)((double*)p + 16);
    }
    if (device < MAX_DEVICES) {
        // the device address of this GPU's copy of `scratch`, looked up once per device with that device current
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[device]) {
            void* ptr = nullptr;
            e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) {
                (void)hipFree(ctx->d_bal);
                (void)hipStreamDestroy(ctx->stream);
                delete ctx;
                return fail(nullptr, RT_EHIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
            }
            g_scratch_ptr[device] = ptr;
        }
        ctx->scratch = g_scratch_ptr[device];
        e = hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipFree(ctx->d_bal);
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail(nullptr, RT_EHIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
        }
    }
    *out = ctx;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)rt_comm_destroy(ctx);
    free_mlp(ctx->mlp[0]);
    free_mlp(ctx->mlp[1]);
    DevBuf* bufs[] = {&ctx->b_y, &ctx->b_xhat, &ctx->b_xs, &ctx->b_out, &ctx->b_partial, &ctx->b_in, &ctx->b_fout,
                      &ctx->b_csend, &ctx->b_crecv, &ctx->b_sum, &ctx->b_mask};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    if (ctx->d_bal) {
        double cum[9];
        if (ctx->balance && ctx->device < MAX_DEVICES &&
            hipMemcpy(cum, ctx->d_bal, sizeof cum, hipMemcpyDeviceToHost) == hipSuccess && cum[0] == 0.0 && cum[8] == 1.0) {
            std::lock_guard<std::mutex> lk(g_scratch_mu);
            memcpy(g_bal_seed[ctx->device], cum, sizeof cum);
            g_bal_seeded[ctx->device] = true;
        }
        (void)hipFree(ctx->d_bal);
    }
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->scratch_done) {
        std::lock_guard<std::mutex> lk(g_scratch_mu);          // (the stream has been drained above)
        if (ctx->device < MAX_DEVICES && g_scratch_owner[ctx->device] == ctx) g_scratch_owner[ctx->device] = nullptr;
        (void)hipEventDestroy(ctx->scratch_done);
    }
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
    // the kernels run at one of five widths; narrower / unequal hidden layers are zero-padded up (exact)
    int hid = hmax <= 128 ? 128 : hmax <= 256 ? 256 : hmax <= 512 ? 512 : 1024;
    const int nst = hid == 1024 ? 1 : 2;
    if (dims[0] > MAX_IN) return fail(ctx, RT_EUNSUPPORTED, "input dim > 128");
    if (dims[n_layers] > MAX_OUT) return fail(ctx, RT_EUNSUPPORTED, "output dim > 128");
    // narrow nets run as one-wave workgroups, four to a CU (rt_mlp.hpp, NW = 1), at width 64 or 128: needs the output
    // layer in one 32-feature tile and a quarter of the LDS (40 KiB) per workgroup.  RT_IS_NARROW=0 (read here) keeps
    // them on the four-wave kernel, zero-padded to width 128 — the same bits, for A/B timing.
    // Width 256 likewise as two-wave workgroups, two to a CU (half the LDS each).
    int nw = 4;
    if (hmax <= 256 && dims[n_layers] <= 32) {
        const int h1 = hmax <= 64 ? 64 : hmax <= 128 ? 128 : 256;
        const int nw1 = h1 == 256 ? 2 : 1;
        const char* env = getenv("RT_IS_NARROW");
        if (!(env && env[0] == '0') &&
            carve_for(h1, 64, ((dims[0] + 7) / 8) * 8, dims[0], dims[n_layers], 64 * nw1).total <= LDS_MAX / (4 / nw1)) {
            nw = nw1;
            hid = h1;
        }
    }
    {
        const int need = carve_for(hid, 32 * nst, ((dims[0] + 7) / 8) * 8, dims[0], dims[n_layers], 64 * nw).total;
        if (need > LDS_MAX) {
            char msg[256];
            snprintf(msg, sizeof msg, "input dim %d with output dim %d at hidden width %d needs %d bytes of LDS per pass (input "
                     "image + the larger of the activation image and the output layer's partial sums); the CU has %d",
                     dims[0], dims[n_layers], hid, need, LDS_MAX);
            return fail(ctx, RT_EUNSUPPORTED, msg);
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HostMlp& M = ctx->mlp[which];
    free_mlp(M);
    M.n_layers = n_layers;
    M.hid = hid;
    M.nst = nst;
    M.nw = nw;
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
    {   // the tangent passes' view of the net: same weights, zero biases on the hidden layers
        void* dz = nullptr;
        const size_t zb = (size_t)((hid + 31) / 32) * 32 * sizeof(float);
        HIPCHK(ctx, hipMalloc(&dz, zb));
        M.dev.push_back(dz);
        HIPCHK(ctx, hipMemset(dz, 0, zb));
        M.args0 = M.args;
        for (int l = 0; l + 1 < n_layers; ++l) M.args0.bp[l] = (const float*)dz;
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

int rt_cf_estimate_dev(rt_ctx* ctx, const double* y_dev, int64_t N, double sd_y, int log_flag, double* out_dev) {
    if (!ctx) return RT_EINVAL;
    if (N > 0 && (!y_dev || !out_dev)) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return cf_enqueue(ctx, y_dev, N, sd_y, log_flag, out_dev);
}

int rt_cf_estimate(rt_ctx* ctx, const double* y, int64_t N, double sd_y, int log_flag, double* out) {
    if (!ctx) return RT_EINVAL;
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "the Laplace estimator needs both G (RT_MLP_G) and H (RT_MLP_H)");
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if (N > 0 && (!y || !out)) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t yb = (size_t)N * G.args.out_dim * sizeof(double), ob = (size_t)N * sizeof(double);
    if (N > 0) {
        int rc = ensure(ctx, ctx->b_y, yb); if (rc) return rc;
        rc = ensure(ctx, ctx->b_out, ob); if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->b_y.p, y, yb, hipMemcpyHostToDevice, ctx->stream));
    }
    const int rc = cf_enqueue(ctx, (const double*)ctx->b_y.p, N, sd_y, log_flag, (double*)ctx->b_out.p);
    if (rc || N == 0) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_out.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_mean_stderr_dev(rt_ctx* ctx, const double* values_dev, int64_t N, double* out2_dev) {
    if (!ctx) return RT_EINVAL;
    if ((N > 0 && !values_dev) || !out2_dev) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return mean_stderr_enqueue(ctx, values_dev, N, out2_dev);
}

int rt_mean_stderr(rt_ctx* ctx, const double* values, int64_t N, double* mean, double* stderr_out) {
    if (!ctx) return RT_EINVAL;
    if (N < 0) return fail(ctx, RT_EINVAL, "N < 0");
    if ((N > 0 && !values) || !mean || !stderr_out) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->b_out, (size_t)(N > 0 ? N : 1) * sizeof(double)); if (rc) return rc;
    if (N > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->b_out.p, values, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = mean_stderr_enqueue(ctx, (const double*)ctx->b_out.p, N, nullptr); if (rc) return rc;
    double res[2];
    const int64_t nchunks = (N + NP_BUF - 1) / NP_BUF;
    HIPCHK(ctx, hipMemcpyAsync(res, (const double*)ctx->b_sum.p + nchunks, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *mean = res[0];
    *stderr_out = res[1];
    return RT_OK;
}

static int is_host(rt_ctx* ctx, const double* y, const float* xhat, const double* xs, int64_t N,
                   int64_t y_index_base, int64_t S, int n_sd, const double* sd_ys, double scale, double df,
                   uint64_t seed, int log_flag, double* out, bool given, double* mean_se = nullptr) {
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
    if (mean_se) {   // the statistic of the results where they are: no second trip over the bus (n_sd == 1 here)
        rc = mean_stderr_enqueue(ctx, (const double*)ctx->b_out.p, N, nullptr); if (rc) return rc;
        const int64_t nchunks = (N + NP_BUF - 1) / NP_BUF;
        HIPCHK(ctx, hipMemcpyAsync(mean_se, (const double*)ctx->b_sum.p + nchunks, 2 * sizeof(double), hipMemcpyDeviceToHost,
                                   ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_is_estimate_summary(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base,
                           int64_t S, double sd_y, double scale, double df, uint64_t seed, double* out,
                           double* mean, double* stderr_out) {
    if (!ctx) return RT_EINVAL;
    if (!mean || !stderr_out) return fail(ctx, RT_EINVAL, "NULL argument");
    if (N <= 0) return fail(ctx, RT_EINVAL, "rt_is_estimate_summary needs N >= 1");
    double ms[2];
    const int rc = is_host(ctx, y, xhat, nullptr, N, y_index_base, S, 1, &sd_y, scale, df, seed, 1, out, false, ms);
    if (rc) return rc;
    *mean = ms[0];
    *stderr_out = ms[1];
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

int rt_is_estimate_dev_noalloc(rt_ctx* ctx, const double* y_dev, const float* xhat_dev, int64_t N,
                               int64_t y_index_base, int64_t S, double sd_y, double scale, double df,
                               uint64_t seed, int log_flag, double* out_dev) {
    if (!ctx) return RT_EINVAL;
    if (N > 0 && (!y_dev || !xhat_dev || !out_dev)) return fail(ctx, RT_EINVAL, "NULL argument");
    if (!ctx->scratch) return fail(ctx, RT_EUNSUPPORTED, "no scratch array on this device (HIP device index >= 64)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const PairBuf pb{(double*)ctx->scratch, (size_t)RT_NOALLOC_SCRATCH_BYTES, false};
    // one context at a time per GPU: the array is shared.  Calls of ONE context are ordered by its stream; another
    // context's call is refused while the last user's work on the array is still outstanding (no allocation: the event
    // is the owner's own, made with its context).  The lock covers enqueue + record, so the hand-over cannot interleave.
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    rt_ctx* owner = g_scratch_owner[ctx->device];
    if (owner && owner != ctx) {
        const hipError_t q = hipEventQuery(g_scratch_event[ctx->device]);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            return fail(ctx, RT_EBUSY, "rt_is_estimate_dev_noalloc: another context's call on this GPU's scratch array is "
                                       "still outstanding (the array is shared per GPU: serialise the callers, or rt_sync the other context first)");
        }
        if (q != hipSuccess) return fail(ctx, RT_EHIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
    }
    g_scratch_owner[ctx->device] = ctx;
    g_scratch_event[ctx->device] = ctx->scratch_done;
    const int rc = is_enqueue(ctx, y_dev, xhat_dev, nullptr, N, y_index_base, S, 1, &sd_y, scale, df, seed, log_flag, out_dev, &pb);
    HIPCHK(ctx, hipEventRecord(ctx->scratch_done, ctx->stream));
    return rc;
}

int rt_is_kernel_name(rt_ctx* ctx, double df, char* buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return RT_EINVAL;
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set");
    SamplerArgs sa;
    const int rc = make_sampler(ctx, 1.0, df, 0, sa);
    if (rc) return rc;
    const bool share = G.nw > 1 && sa.ndf + G.args.in_dim > 4;      // launch_is's choice
    snprintf(buf, cap, "is_fused_kernel<%d, %d, false, %s, %d>", G.hid, G.nst, share ? "true" : "false", G.nw);
    return RT_OK;
}

int rt_is_pair_count(rt_ctx* ctx, int64_t S, int64_t* E) {
    if (!ctx || !E) return RT_EINVAL;
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set (its width fixes the samples per pass)");
    if (S <= 0 || S > 0xFFFFFFFFll) return fail(ctx, RT_EINVAL, "need 0 < S < 2^32");
    *E = is_plan(1, S, G.nst).E;
    return RT_OK;
}

int rt_is_pairs(rt_ctx* ctx, const double* y, const float* xhat, int64_t N, int64_t y_index_base, int64_t S,
                int n_sd, const double* sd_y, double scale, double df, uint64_t seed, int64_t e_lo, int64_t e_hi,
                double* pairs) {
    if (!ctx) return RT_EINVAL;
    HostMlp& G = ctx->mlp[RT_MLP_G];
    if (!G.set) return fail(ctx, RT_ENOTSET, "G (RT_MLP_G) has not been set");
    if (N < 0 || S <= 0 || S > 0xFFFFFFFFll) return fail(ctx, RT_EINVAL, "need N >= 0 and 0 < S < 2^32");
    if (n_sd < 1 || n_sd > RT_MAX_SD || !sd_y) return fail(ctx, RT_EINVAL, "n_sd must be in [1, RT_MAX_SD]");
    for (int k = 0; k < n_sd; ++k)
        if (!(sd_y[k] > 0.0) || !isfinite(sd_y[k])) return fail(ctx, RT_EINVAL, "sd_y must be positive and finite");
    const int64_t E = is_plan(1, S, G.nst).E;
    if (e_lo < 0 || e_hi < e_lo || e_hi > E) return fail(ctx, RT_EINVAL, "need 0 <= e_lo <= e_hi <= rt_is_pair_count(S)");
    const int64_t cnt = e_hi - e_lo;
    if (N == 0 || cnt == 0) return RT_OK;
    if (!y || !xhat || !pairs) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int dx = G.args.in_dim, dy = G.args.out_dim, MS = 32 * G.nst;
    const size_t yb = (size_t)N * dy * sizeof(double), xb = (size_t)N * dx * sizeof(float);
    int rc = ensure(ctx, ctx->b_y, yb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_xhat, xb); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_y.p, y, yb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_xhat.p, xhat, xb, hipMemcpyHostToDevice, ctx->stream));
    // blocks of query points, bounded like rt_is_estimate's launches (sample-evals of THIS range) and to 1 GiB of pairs
    int64_t per = ctx->max_launch_evals / (cnt * MS);
    const int64_t per_mem = (int64_t)((1ull << 30) / ((size_t)E * n_sd * 2 * sizeof(double)));
    if (per > per_mem) per = per_mem;
    if (per < 1) per = 1;
    if (per > N) per = N;
    rc = ensure(ctx, ctx->b_partial, (size_t)per * E * n_sd * 2 * sizeof(double));
    if (rc) return rc;
    const size_t entry = (size_t)n_sd * 2 * sizeof(double);
    for (int64_t n0 = 0; n0 < N; n0 += per) {
        const int64_t nb = (N - n0 < per) ? N - n0 : per;
        rc = is_enqueue_block(ctx, PairBuf{(double*)ctx->b_partial.p, ctx->b_partial.cap, true},
                              (const double*)ctx->b_y.p + n0 * dy, (const float*)ctx->b_xhat.p + n0 * dx, nullptr, nb,
                              y_index_base + n0, S, n_sd, sd_y, scale, df, seed, 1, nullptr, 0, e_lo, e_hi);
        if (rc) return rc;
        // rows = query points, row = this range's cnt entries out of the E the device buffer holds per point
        HIPCHK(ctx, hipMemcpy2DAsync(pairs + (size_t)n0 * cnt * n_sd * 2, cnt * entry,
                                     (const char*)ctx->b_partial.p + e_lo * entry, E * entry, cnt * entry, nb,
                                     hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // the next block reuses the device buffer
    }
    return RT_OK;
}

int rt_is_merge_pairs(rt_ctx* ctx, const double* pairs, int64_t N, int64_t E, int n_sd, int64_t S, int log_flag,
                      double* out) {
    if (!ctx) return RT_EINVAL;
    if (N < 0 || E < 1 || S <= 0) return fail(ctx, RT_EINVAL, "need N >= 0, E >= 1, S > 0");
    if (n_sd < 1 || n_sd > RT_MAX_SD) return fail(ctx, RT_EINVAL, "n_sd must be in [1, RT_MAX_SD]");
    if (N == 0) return RT_OK;
    if (!pairs || !out) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t pb = (size_t)N * E * n_sd * 2 * sizeof(double), ob = (size_t)N * n_sd * sizeof(double);
    int rc = ensure(ctx, ctx->b_partial, pb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_out, ob); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_partial.p, pairs, pb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(is_finalize_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const double*)ctx->b_partial.p, N, E, n_sd, log((double)S), log_flag, (double*)ctx->b_out.p, N);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_out.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
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

int rt_philox4x32_10(rt_ctx* ctx, const uint32_t* ctr_key, int64_t n, uint32_t* out) {
    if (!ctx) return RT_EINVAL;
    if (n < 0) return fail(ctx, RT_EINVAL, "n < 0");
    if (n == 0) return RT_OK;
    if (!ctr_key || !out) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t ib = (size_t)n * 6 * sizeof(uint32_t), ob = (size_t)n * 4 * sizeof(uint32_t);
    int rc = ensure(ctx, ctx->b_in, ib); if (rc) return rc;
    rc = ensure(ctx, ctx->b_fout, ob); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_in.p, ctr_key, ib, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)ctx->b_in.p, n, (uint32_t*)ctx->b_fout.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_fout.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_log1p_pos(rt_ctx* ctx, const double* a, int64_t n, double* out) {
    if (!ctx) return RT_EINVAL;
    if (n < 0) return fail(ctx, RT_EINVAL, "n < 0");
    if (n == 0) return RT_OK;
    if (!a || !out) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nb = (size_t)n * sizeof(double);
    int rc = ensure(ctx, ctx->b_in, nb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_fout, nb); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_in.p, a, nb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(log1p_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)ctx->b_in.p, n, (double*)ctx->b_fout.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_fout.p, nb, hipMemcpyDeviceToHost, ctx->stream));
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
    const int rc = harvest_events(ctx);
    if (rc) return rc;
    out->kernel_ms = ctx->ms_harvested;
    out->flops = ctx->flops_acc;
    out->launches = ctx->launches_harvested;
    out->sample_evals = ctx->evals_acc;
    return RT_OK;
}

int rt_reset_counters(rt_ctx* ctx) {
    if (!ctx) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->events_used = 0;
    ctx->ms_harvested = 0.0;
    ctx->launches_harvested = 0;
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

#if defined(RT_STAMPS) && RT_STAMPS == 2
int rt_debug_wg_ticks(rt_ctx* ctx, unsigned long long* out2048) {
    if (!ctx || !out2048) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_wg_ticks), sizeof(unsigned long long) * 2048));
    return RT_OK;
}
#endif

int rt_device_cu_count(rt_ctx* ctx, int* out) {
    if (!ctx || !out) return RT_EINVAL;
    *out = ctx->cu_count;
    return RT_OK;
}

int rt_get_xcd_shares(rt_ctx* ctx, double* shares8) {
    if (!ctx || !shares8) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double cum[9];
    HIPCHK(ctx, hipMemcpy(cum, ctx->d_bal, sizeof cum, hipMemcpyDeviceToHost));
    for (int x = 0; x < 8; ++x) shares8[x] = cum[x + 1] - cum[x];
    return RT_OK;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing by 8: host code for the checkpoint reader.
uint32_t rt_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
    });
    const unsigned char* p = (const unsigned char*)data;
    crc = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
              T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

// ---- device memory for callers of the *_dev entry points ---------------------------------------------------------
int rt_dev_alloc(rt_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return RT_EINVAL;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(out, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(ctx, RT_ENOMEM, "hipMalloc failed");
    }
    return RT_OK;
}

int rt_dev_mem_info(rt_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return RT_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return RT_OK;
}

int rt_dev_free(rt_ctx* ctx, void* ptr) {
    if (!ctx) return RT_EINVAL;
    if (!ptr) return RT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // nothing enqueued may still use it
    HIPCHK(ctx, hipFree(ptr));
    return RT_OK;
}

int rt_memcpy_h2d(rt_ctx* ctx, void* dst_dev, const void* src, size_t bytes) {
    if (!ctx) return RT_EINVAL;
    if (bytes == 0) return RT_OK;
    if (!dst_dev || !src) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_memcpy_d2h(rt_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
    if (!ctx) return RT_EINVAL;
    if (bytes == 0) return RT_OK;
    if (!dst || !src_dev) return fail(ctx, RT_EINVAL, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ---- the one collective of the path: RCCL, on the context's stream -----------------------------------------------
// librccl.so.1 is loaded on first use (a one-GPU process never pays for it) and its entry points are taken from that
// handle, so whichever RCCL the process already holds under that soname is the one used.
namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

const Rccl* rccl_load(std::string& err) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return &g_rccl;
    if (!g_rccl.err.empty()) { err = g_rccl.err; return nullptr; }
    const char* names[] = {getenv("RT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    std::string tried;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(n) + ": " + dlerror() + "; ";
    }
    if (!h) { g_rccl.err = err = "cannot load RCCL (" + tried + ")"; return nullptr; }
    bool ok = true;
    auto sym = [&](const char* n) { void* f = dlsym(h, n); if (!f) { ok = false; tried += std::string("missing ") + n + "; "; } return f; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(h); g_rccl.err = err = "RCCL library lacks entry points (" + tried + ")"; return nullptr; }
    g_rccl.handle = h;
    return &g_rccl;
}

#define NCCLCHK(ctx, R, call)                                                                         \
    do {                                                                                              \
        const ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) return fail(ctx, RT_ECOMM, std::string(#call) + ": " + (R)->GetErrorString(r_)); \
    } while (0)
}  // namespace

int rt_comm_unique_id(void* id128) {
    if (!id128) return fail(nullptr, RT_EINVAL, "id128 is NULL");
    std::string err;
    const Rccl* R = rccl_load(err);
    if (!R) return fail(nullptr, RT_ECOMM, err);
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES must be RCCL's unique-id size");
    ncclUniqueId id;
    NCCLCHK(nullptr, R, R->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return RT_OK;
}

int rt_comm_init(rt_ctx* ctx, int nranks, int rank, const void* id128) {
    if (!ctx) return RT_EINVAL;
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, RT_EINVAL, "need id128, 0 <= rank < nranks");
    if (ctx->comm) return fail(ctx, RT_EINVAL, "this context already has a communicator (rt_comm_destroy first)");
    std::string err;
    const Rccl* R = rccl_load(err);
    if (!R) return fail(ctx, RT_ECOMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(ctx, R, R->CommInitRank(&ctx->comm, nranks, id, rank));
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return RT_OK;
}

int rt_comm_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_EINVAL;
    if (!ctx->comm) return RT_OK;
    std::string err;
    const Rccl* R = rccl_load(err);
    if (!R) return fail(ctx, RT_ECOMM, err);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const ncclComm_t c = ctx->comm;
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_nranks = 1;
    NCCLCHK(ctx, R, R->CommDestroy(c));
    return RT_OK;
}

int rt_comm_info(rt_ctx* ctx, int* rank, int* nranks) {
    if (!ctx) return RT_EINVAL;
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm_nranks;
    return RT_OK;
}

int rt_comm_all_gather_f64_dev(rt_ctx* ctx, const double* send_dev, int64_t count, double* recv_dev) {
    if (!ctx) return RT_EINVAL;
    if (!ctx->comm) return fail(ctx, RT_ENOTSET, "rt_comm_init has not been called on this context");
    if (count < 0 || (count > 0 && (!send_dev || !recv_dev))) return fail(ctx, RT_EINVAL, "bad count or NULL argument");
    if (count == 0) return RT_OK;
    std::string err;
    const Rccl* R = rccl_load(err);
    if (!R) return fail(ctx, RT_ECOMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    NCCLCHK(ctx, R, R->AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, ctx->comm, ctx->stream));
    return RT_OK;
}

int rt_comm_all_gather_f64(rt_ctx* ctx, const double* send, int64_t count, double* recv) {
    if (!ctx) return RT_EINVAL;
    if (!ctx->comm) return fail(ctx, RT_ENOTSET, "rt_comm_init has not been called on this context");
    if (count < 0 || (count > 0 && (!send || !recv))) return fail(ctx, RT_EINVAL, "bad count or NULL argument");
    if (count == 0) return RT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t sb = (size_t)count * sizeof(double), rb = sb * (size_t)ctx->comm_nranks;
    int rc = ensure(ctx, ctx->b_csend, sb); if (rc) return rc;
    rc = ensure(ctx, ctx->b_crecv, rb); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_csend.p, send, sb, hipMemcpyHostToDevice, ctx->stream));
    rc = rt_comm_all_gather_f64_dev(ctx, (const double*)ctx->b_csend.p, count, (double*)ctx->b_crecv.p);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(recv, ctx->b_crecv.p, rb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_comm_all_reduce_max_f64(rt_ctx* ctx, double* values, int64_t count) {
    if (!ctx) return RT_EINVAL;
    if (!ctx->comm) return fail(ctx, RT_ENOTSET, "rt_comm_init has not been called on this context");
    if (count < 0 || (count > 0 && !values)) return fail(ctx, RT_EINVAL, "bad count or NULL argument");
    if (count == 0) return RT_OK;
    std::string err;
    const Rccl* R = rccl_load(err);
    if (!R) return fail(ctx, RT_ECOMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t sb = (size_t)count * sizeof(double);
    int rc = ensure(ctx, ctx->b_csend, sb); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_csend.p, values, sb, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ctx, R, R->AllReduce(ctx->b_csend.p, ctx->b_csend.p, (size_t)count, ncclDouble, ncclMax, ctx->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(values, ctx->b_csend.p, sb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_comm_barrier(rt_ctx* ctx) {
    double one = 1.0;
    return rt_comm_all_reduce_max_f64(ctx, &one, 1);
}

}  // extern "C"

