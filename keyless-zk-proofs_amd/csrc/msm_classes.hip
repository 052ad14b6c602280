// msm_classes.hip -- classification of an MSM's scalar vector by size class (group-independent part of the scalar-class
// MSM; the point arithmetic is in msm_kernels.inc, "Scalar-class MSM").
//
// Replaces the zero-digit skip of the reference's bucket loop (rust-rapidsnark/rapidsnark/src/multiexp.cpp:59-65,
// `if (chunkValue)`), which is what makes the four witness MSMs of a proof (groth16.cpp:88-112) cheap on a CPU: the
// scalars are the circuit's wire values, mostly 0 and 1, some bytes, few field elements.  One pass over the scalars emits
//   * for every mask set s and bit b < 8: the list of wires i with scalar_i < 256, bit b of scalar_i set, and row i of the
//     set's table not (0,0)  (a table's (0,0) rows -- wires a constraint side never mentions -- add nothing:
//     curve.cpp:185-250 returns the other operand);
//   * the scalars >= 256 ("wide"), compacted, with their wire numbers.
// A, B1, B2 and C are all indexed by wire (the prover stores C with n_public + 1 leading (0,0) rows), so one classification
// serves the four MSMs.
#include <string.h>
#include <algorithm>
#include "ctx.h"

namespace {

constexpr int      MAX_SETS = k16_scalar_classes::MAX_SETS;
constexpr int      BITS     = k16_scalar_classes::BITS;
constexpr int      WIDE_CNT = MAX_SETS * BITS; // index of the wide counter in d_cnt
constexpr unsigned CLS_PPT  = 8;               // scalars per lane
constexpr unsigned CLS_TILE = 256 * CLS_PPT;

struct MaskPtrs {
    const uint64_t* m[MAX_SETS]; // bit i set: row i of the set's table is (0,0); nullptr: no row is
};

// bit i of mask = (row i is all zero); rows of ROW_BYTES (64: G1, 128: G2), one lane per row, one 64-bit word per wave
template <unsigned ROW_BYTES>
__global__ void __launch_bounds__(256) k_zero_row_mask(const uint4* __restrict__ rows, uint64_t n, uint64_t* __restrict__ mask)
{
    const uint64_t i    = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool           zero = false;
    if (i < n) {
        uint32_t any = 0;
#pragma unroll
        for (unsigned k = 0; k < ROW_BYTES / 16; k++) {
            const uint4 v = rows[i * (ROW_BYTES / 16) + k];
            any |= v.x | v.y | v.z | v.w;
        }
        zero = any == 0;
    }
    const uint64_t m = __ballot(zero);
    if ((threadIdx.x & 63u) == 0 && (i >> 6) < (n + 63) / 64) mask[i >> 6] = m;
}

// One workgroup classifies a tile of 2048 scalars (lane l of wave w takes scalar tile * 2048 + p * 256 + w * 64 + l in
// trip p, so a ballot is 64 consecutive wires).  Phase 1 counts every list's entries of the tile, one global atomic per
// list reserves the tile's run, phase 2 recomputes the ballots and writes the wire numbers -- in index order inside a
// tile, tiles in the order of their reservations.
template <int NSETS>
__global__ void __launch_bounds__(256) k_classify(const uint4* __restrict__ scalars, uint32_t n, MaskPtrs masks,
                                                  uint32_t* __restrict__ cnt, uint32_t* __restrict__ lists, uint64_t cap_n,
                                                  uint32_t* __restrict__ wide_idx, uint4* __restrict__ wide_scalars,
                                                  uint32_t wide_cap, uint32_t* __restrict__ flags)
{
    constexpr int       NL = NSETS * BITS + 1; // lists + the wide list
    __shared__ uint32_t wave_cnt[4][NL];
    __shared__ uint32_t wave_base[4][NL];
    const uint32_t      wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t      t0   = blockIdx.x * CLS_TILE;
    // code: low byte | wide << 8 | (row of set s is (0,0)) << (9 + s) | out of range << 15
    uint32_t code[CLS_PPT];
#pragma unroll
    for (unsigned p = 0; p < CLS_PPT; p++) {
        const uint32_t i = t0 + p * 256 + threadIdx.x;
        uint32_t       c = 1u << 15;
        if (i < n) {
            const uint4 lo = scalars[2 * (uint64_t)i], hi = scalars[2 * (uint64_t)i + 1];
            const bool  wide = ((lo.x >> 8) | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0;
            c = wide ? (1u << 8) : (lo.x & 0xffu);
#pragma unroll
            for (int s = 0; s < NSETS; s++)
                if (masks.m[s] && ((masks.m[s][i >> 6] >> (i & 63u)) & 1ull)) c |= 1u << (9 + s);
        }
        code[p] = c;
    }
    auto pred = [&](uint32_t c, int s, int b) -> bool { return !(c & ((1u << 15) | (1u << 8) | (1u << (9 + s)))) && ((c >> b) & 1u); };
    auto is_wide = [&](uint32_t c) -> bool { return (c & ((1u << 15) | (1u << 8))) == (1u << 8); };
    // phase 1: this wave's count of every list
    uint32_t mine = 0; // lane q < NL ends up holding the wave's count of list q
#pragma unroll
    for (int s = 0; s < NSETS; s++)
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            uint32_t k = 0;
#pragma unroll
            for (unsigned p = 0; p < CLS_PPT; p++) k += (uint32_t)__popcll(__ballot(pred(code[p], s, b)));
            if ((int)lane == s * BITS + b) mine = k;
        }
    {
        uint32_t k = 0;
#pragma unroll
        for (unsigned p = 0; p < CLS_PPT; p++) k += (uint32_t)__popcll(__ballot(is_wide(code[p])));
        if ((int)lane == NL - 1) mine = k;
    }
    if ((int)lane < NL) wave_cnt[wave][lane] = mine;
    __syncthreads();
    if ((int)threadIdx.x < NL) {
        const int      q   = (int)threadIdx.x;
        const uint32_t c0 = wave_cnt[0][q], c1 = wave_cnt[1][q], c2 = wave_cnt[2][q], c3 = wave_cnt[3][q];
        const uint32_t tot = c0 + c1 + c2 + c3;
        const int      gq  = q == NL - 1 ? WIDE_CNT : q;
        const uint32_t g   = tot ? atomicAdd(&cnt[gq], tot) : 0u;
        wave_base[0][q] = g;
        wave_base[1][q] = g + c0;
        wave_base[2][q] = g + c0 + c1;
        wave_base[3][q] = g + c0 + c1 + c2;
    }
    __syncthreads();
    // phase 2
#pragma unroll
    for (int s = 0; s < NSETS; s++)
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            uint32_t  run = wave_base[wave][s * BITS + b];
            uint32_t* out = lists + (size_t)(s * BITS + b) * cap_n;
#pragma unroll
            for (unsigned p = 0; p < CLS_PPT; p++) {
                const bool     on = pred(code[p], s, b);
                const uint64_t m  = __ballot(on);
                if (on) out[run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = t0 + p * 256 + threadIdx.x;
                run += (uint32_t)__popcll(m);
            }
        }
    {
        uint32_t run = wave_base[wave][NL - 1];
#pragma unroll
        for (unsigned p = 0; p < CLS_PPT; p++) {
            const bool     on = is_wide(code[p]);
            const uint64_t m  = __ballot(on);
            if (on) {
                const uint32_t i = t0 + p * 256 + threadIdx.x;
                const uint32_t k = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (k < wide_cap) {
                    wide_idx[k]             = i;
                    wide_scalars[2 * (uint64_t)k]     = scalars[2 * (uint64_t)i];
                    wide_scalars[2 * (uint64_t)k + 1] = scalars[2 * (uint64_t)i + 1];
                } else {
                    flags[0] = 1u; // more wide scalars than the caller announced: the MSMs built on this must fail
                }
            }
            run += (uint32_t)__popcll(m);
        }
    }
}

__global__ void k_publish_counts(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ host)
{
    if (threadIdx.x <= WIDE_CNT) host[1 + threadIdx.x] = cnt[threadIdx.x];
}

void classes_free(k16_scalar_classes* c)
{
    if (!c) return;
    if (c->d_cnt) (void)hipFree(c->d_cnt);
    if (c->d_lists) (void)hipFree(c->d_lists);
    if (c->d_wide_idx) (void)hipFree(c->d_wide_idx);
    if (c->d_wide_scalars) (void)hipFree(c->d_wide_scalars);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->built) (void)hipEventDestroy(c->built);
    delete c;
}

} // namespace

extern "C" int k16_scalar_classes_create(k16_ctx* ctx, uint64_t max_n, int max_sets, k16_scalar_classes** out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !out || max_sets < 1 || max_sets > MAX_SETS || max_n >= (1ull << 31)) return K16_ERR_ARG;
    *out = nullptr;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    k16_scalar_classes* c = new k16_scalar_classes();
    c->ctx      = ctx;
    c->cap_n    = std::max<uint64_t>(max_n, 1);
    c->max_sets = max_sets;
    if (hipMalloc((void**)&c->d_cnt, (WIDE_CNT + 8) * 4) != hipSuccess ||
        hipMalloc((void**)&c->d_lists, (size_t)max_sets * BITS * c->cap_n * 4) != hipSuccess ||
        hipMalloc((void**)&c->d_wide_idx, c->cap_n * 4) != hipSuccess ||
        hipMalloc((void**)&c->d_wide_scalars, c->cap_n * 32) != hipSuccess ||
        hipHostMalloc((void**)&c->h_flags, 64 * 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->h_flags_dev, c->h_flags, 0) != hipSuccess ||
        hipEventCreateWithFlags(&c->built, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        classes_free(c);
        ctx->err = "k16_scalar_classes_create: device allocation failed";
        return K16_ERR_HIP;
    }
    memset(c->h_flags, 0, 64 * 4);
    *out = c;
    return K16_OK;
    });
}

extern "C" void k16_scalar_classes_destroy(k16_scalar_classes* c)
{
    k16_guard_void([&]() {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipDeviceSynchronize();
    // MSMs still waiting for their k16_msm_finish keep a pointer to this object for its overflow flag (written by the
    // device, final after the synchronisation above): hand them the flag's value instead (use-after-free)
    {
        std::lock_guard<std::mutex> lk(c->ctx->ring_mu);
        for (auto& pd : c->ctx->pend)
            if (pd.cls == c) {
                pd.cls_overflow = c->h_flags && c->h_flags[0] != 0;
                pd.cls          = nullptr;
            }
    }
    classes_free(c);
    });
}

extern "C" int k16_msm_zero_row_mask(k16_ctx* ctx, int group, const void* d_rows, uint64_t n, void* d_mask)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || (group != K16_G1 && group != K16_G2) || (n && (!d_rows || !d_mask))) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (group == K16_G1)
        hipLaunchKernelGGL(k_zero_row_mask<64>, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)d_rows, n, (uint64_t*)d_mask);
    else
        hipLaunchKernelGGL(k_zero_row_mask<128>, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)d_rows, n, (uint64_t*)d_mask);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
    });
}

// Runs on the CURRENT lane's stream.  n_wide_bound < 0: the call waits for the classification and reads the number of wide
// scalars back (exact); >= 0: the caller's upper bound on it (the prover counts them while it packs the witness) -- nothing
// waits, the wide MSM runs over that many rows (unused ones hold zero scalars), and an actual count above the bound makes
// every MSM enqueued from these classes fail at k16_msm_finish.
extern "C" int k16_scalar_classes_build(k16_ctx* ctx, k16_scalar_classes* c, const void* d_scalars, uint64_t n,
                                        const void* const* d_zero_masks, int n_sets, int64_t n_wide_bound)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !c || c->ctx != ctx || n > c->cap_n || n_sets < 1 || n_sets > c->max_sets || (n && !d_scalars) ||
        n_wide_bound > (int64_t)n)
        return K16_ERR_ARG;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = k16_lane_stream(ctx, ctx->cur_lane);
    c->n      = n;
    c->n_sets = n_sets;
    c->n_wide = 0;
    c->h_flags[0] = 0;
    if (n == 0) {
        K16_HIP(ctx, hipEventRecord(c->built, st));
        return K16_OK;
    }
    const uint64_t wide_cap = n_wide_bound < 0 ? n : (uint64_t)n_wide_bound;
    MaskPtrs       mp;
    for (int s = 0; s < MAX_SETS; s++) mp.m[s] = (d_zero_masks && s < n_sets) ? (const uint64_t*)d_zero_masks[s] : nullptr;
    K16_HIP(ctx, hipMemsetAsync(c->d_cnt, 0, (WIDE_CNT + 8) * 4, st));
    if (n_wide_bound > 0) K16_HIP(ctx, hipMemsetAsync(c->d_wide_scalars, 0, wide_cap * 32, st)); // rows the count does not reach: zero scalars
    const unsigned grid = (unsigned)((n + CLS_TILE - 1) / CLS_TILE);
#define K16_CLASSIFY(NS)                                                                                                  \
    hipLaunchKernelGGL(k_classify<NS>, dim3(grid), dim3(256), 0, st, (const uint4*)d_scalars, (uint32_t)n, mp, c->d_cnt, \
                       c->d_lists, c->cap_n, c->d_wide_idx, (uint4*)c->d_wide_scalars, (uint32_t)wide_cap, c->h_flags_dev)
    switch (n_sets) {
    case 1: K16_CLASSIFY(1); break;
    case 2: K16_CLASSIFY(2); break;
    case 3: K16_CLASSIFY(3); break;
    default: K16_CLASSIFY(4); break;
    }
#undef K16_CLASSIFY
    K16_HIP(ctx, hipGetLastError());
    if (n_wide_bound < 0) {
        hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(64), 0, st, (const uint32_t*)c->d_cnt, c->h_flags_dev);
        K16_HIP(ctx, hipEventRecord(c->built, st));
        K16_HIP(ctx, hipEventSynchronize(c->built));
        c->n_wide = c->h_flags[1 + WIDE_CNT];
    } else {
        K16_HIP(ctx, hipEventRecord(c->built, st));
        c->n_wide = wide_cap;
    }
    return K16_OK;
    });
}

// counts of the last build (synchronises): out[0 .. n_sets*8) list lengths, out[n_sets*8] wide scalars met
extern "C" int k16_scalar_classes_counts(k16_ctx* ctx, const k16_scalar_classes* c, uint32_t* out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !c || c->ctx != ctx || !out) return K16_ERR_ARG;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t h[WIDE_CNT + 8];
    K16_HIP(ctx, hipEventSynchronize(c->built));
    K16_HIP(ctx, hipMemcpy(h, c->d_cnt, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < c->n_sets * BITS; i++) out[i] = h[i];
    out[c->n_sets * BITS] = h[WIDE_CNT];
    return K16_OK;
    });
}
