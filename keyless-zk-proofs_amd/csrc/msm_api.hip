// msm_api.hip -- C-ABI entry points of the MSM (host side: window choice, Horner combine).
// Kernels live in msm_kernels.inc, instantiated for G1 in msm_g1.hip and for G2 in msm_g2.hip.
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>
#include <functional>
#include "ctx.h"
#include "bn254_fq9.h"

using namespace k16;

int k16_msm_enqueue_g1(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c, int prepared);
int k16_msm_enqueue_g2(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c, int prepared);
int k16_msm_prepare_g1(k16_ctx* ctx, const void* d_bases, uint64_t n, void* d_out, hipStream_t st);
int k16_msm_prepare_g2(k16_ctx* ctx, const void* d_bases, uint64_t n, void* d_out, hipStream_t st);
int k16_msm_fixed_tables_g1(k16_ctx* ctx, const void* d_bases, uint64_t n, unsigned c, unsigned W, void* d_table);
int k16_msm_enqueue_fixed_g1(k16_ctx* ctx, const void* d_table, const void* d_scalars, uint64_t n, unsigned c);
int k16_msm_enqueue_classified_g1(k16_ctx* ctx, const void* d_rows, const k16_scalar_classes* cls, int set, unsigned c, bool* has_wide, int phase);
int k16_msm_enqueue_classified_g2(k16_ctx* ctx, const void* d_rows, const k16_scalar_classes* cls, int set, unsigned c, bool* has_wide, int phase);

namespace {
constexpr unsigned MAX_C = 16;
constexpr unsigned MIN_C = 4;
inline unsigned n_windows(unsigned c) { return (256 + c - 1) / c + 1; } // + carry window of the signed recoding

unsigned choose_c(const k16_ctx* ctx, uint64_t n)
{
    if (ctx->forced_c >= MIN_C && ctx->forced_c <= MAX_C) return ctx->forced_c;
    unsigned lg = 0;
    while ((n >> (lg + 1)) != 0) lg++;
    int c = (int)lg - (lg >= 18 ? 3 : 4); // measured on MI355X: 2^16 -> 12/13, 2^18 -> 15, 2^19 -> 15/16, 2^20.. -> 16
    if (c < (int)MIN_C) c = MIN_C;
    if (c > (int)MAX_C) c = MAX_C;
    return (unsigned)c;
}

// Host tail of the MSM: per window  val = T[nbits] + T[nbits+1] + M * sum_b 2^b T[b]  (msm_kernels.inc, K4),
// then the Horner combine over windows (multiexp.cpp:236-242).
template <class F>
Xyzz<F> window_value_host(const Xyzz<F>* tw, unsigned nbits, unsigned mlog);
template <class F>
void window_values_host(k16_ctx* ctx, const Xyzz<F>* T, unsigned W, unsigned nbits, unsigned mlog, std::vector<Xyzz<F>>* out);
template <class F>
void horner_host(k16_ctx* ctx, const Xyzz<F>* T, unsigned W, unsigned c, unsigned nbits, unsigned mlog, Xyzz<F>* out)
{
    std::vector<Xyzz<F>> val;
    window_values_host<F>(ctx, T, W, nbits, mlog, &val);
    Xyzz<F> r = Xyzz<F>::zero();
    for (int w = (int)W - 1; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) r = pdbl(r);
        r = padd(r, val[w]);
    }
    *out = r;
}

// Fixed-base MSM (one bucket set for all digit positions): the device leaves, per pseudo-window v of `mag` magnitudes,
// the same T[v][...] partial sums as an ordinary window; window v stands for the magnitudes v*mag + 1 .. (v+1)*mag, so
//     result = sum_v val_v + mag * sum_v v * Stot_v        (Stot_v = T[v][nbits+1], the plain sum of the window)
// value of one (pseudo-)window from its partial sums: 2^mlog * sum_b 2^b T[b] + T[nbits] + T[nbits + 1]
template <class F>
Xyzz<F> window_value_host(const Xyzz<F>* tw, unsigned nbits, unsigned mlog)
{
    Xyzz<F> t = Xyzz<F>::zero();
    for (int b = (int)nbits - 1; b >= 0; b--) {
        t = pdbl(t);
        t = padd(t, tw[b]);
    }
    for (unsigned k = 0; k < mlog; k++) t = pdbl(t);
    t = padd(t, tw[nbits]);
    return padd(t, tw[nbits + 1]); // + sum of all X: slot i' carries the weight i' + 1
}
// ... of all windows, on the context's host threads when there are some: the ~30 group operations per window are the
// longest host-side stretch on a proof's critical path (the H MSM's combine: 16 windows, ~0.25 ms serially)
// prep(w), if given, runs first in window w's task (the conversion of that window's partial sums from the kernels' field
// representation: it is as much work as the window's group operations)
static thread_local const std::function<void(unsigned)>* g_window_prep = nullptr;
template <class F>
void window_values_host(k16_ctx* ctx, const Xyzz<F>* T, unsigned W, unsigned nbits, unsigned mlog, std::vector<Xyzz<F>>* out)
{
    out->resize(W);
    k16_host_pool* pool = (W >= 4 && ctx->parallel_combine) ? k16_ctx_pool(ctx) : nullptr;
    const std::function<void(unsigned)>* prep = g_window_prep;
    auto one = [&](unsigned w) {
        if (prep) (*prep)(w);
        (*out)[w] = window_value_host<F>(T + (size_t)w * (nbits + 2), nbits, mlog);
    };
    if (pool)
        pool->run(W, one);
    else
        for (unsigned w = 0; w < W; w++) one(w);
}

template <class F>
void flat_combine_host(k16_ctx* ctx, const Xyzz<F>* T, unsigned Wr, unsigned c, unsigned nbits, unsigned mlog, Xyzz<F>* out)
{
    std::vector<Xyzz<F>> val;
    window_values_host<F>(ctx, T, Wr, nbits, mlog, &val);
    Xyzz<F> total = Xyzz<F>::zero(), run = Xyzz<F>::zero(), wsum = Xyzz<F>::zero();
    for (int v = (int)Wr - 1; v >= 0; v--) {
        const Xyzz<F>* tw = T + (size_t)v * (nbits + 2);
        total = padd(total, val[v]);
        if (v > 0) {                      // sum_v v * Stot_v by running sums, from the top window down
            run  = padd(run, tw[nbits + 1]);
            wsum = padd(wsum, run);
        }
    }
    // mag = 2^(c-1) / Wr = 2^(nbits + mlog)
    for (unsigned k = 0; k < nbits + mlog; k++) wsum = pdbl(wsum);
    (void)c;
    *out = padd(total, wsum);
}

// window size of the fixed-base tables for n points: the largest supported c <= log2(n) (0: n too small, use the
// ordinary MSM); W = ceil(257 / c) tables of n rows.  (Cost model: 10 multiplications per (point, digit) pair + ~49 per
// signed bucket: c = 20 wins from n = 2^20, c = 18 from 2^18, c = 16 from 2^16.)
unsigned fixed_base_c(uint64_t n)
{
    unsigned lg = 0;
    while ((n >> (lg + 1)) != 0) lg++;
    static const unsigned cs[] = {20, 18, 16, 14, 12};
    for (unsigned c : cs)
        if (lg >= c && (uint64_t)n * ((257 + c - 1) / c) < (1ull << 25)) return c;
    return 0;
}

} // namespace

extern "C" int k16_msm_fixed_base_info(uint64_t n, unsigned* c_out, uint64_t* table_rows)
{
    return k16_guard(nullptr, [&]() -> int {
    const unsigned c = fixed_base_c(n);
    if (c_out) *c_out = c;
    if (table_rows) *table_rows = c ? (uint64_t)n * ((257 + c - 1) / c) : 0;
    return K16_OK;
    });
}

// The one-shot requests of the NEXT enqueue (zero-row mask, accumulation mask, row indirection, sort reuse / derive, scalars
// still to be formed) cover exactly one enqueue -- also one that returns early: n == 0, a full result ring, n too large, any
// error before msm_enqueue_t has consumed them (they then applied to the next, unrelated MSM, whose rows were
// silently dropped).  Cleared on every way out.
struct OneShotGuard {
    k16_ctx* c;
    explicit OneShotGuard(k16_ctx* cx) : c(cx) {}
    ~OneShotGuard()
    {
        if (!c) return;
        c->skip_next       = nullptr;
        c->acc_skip_next   = nullptr;
        c->remap_next      = nullptr;
        c->derive_lane     = -1;
        c->reuse_sort      = false;
        c->reuse_sort_lane = -1;
        c->hs_next[0] = c->hs_next[1] = c->hs_next[2] = nullptr;
    }
};

extern "C" int k16_msm_fixed_base_prepare(k16_ctx* ctx, int group, const void* d_bases, uint64_t n, void* d_table)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || group != K16_G1 || !d_bases || !d_table) return K16_ERR_ARG;
    const unsigned c = fixed_base_c(n);
    if (!c) {
        ctx->err = "fixed-base msm: table too small or too large for this mode (see k16_msm_fixed_base_info)";
        return K16_ERR_ARG;
    }
    return k16_msm_fixed_tables_g1(ctx, d_bases, n, c, (257 + c - 1) / c, d_table);
    });
}

extern "C" int k16_msm_enqueue_fixed_base(k16_ctx* ctx, int group, const void* d_table, const void* d_scalars, uint64_t n)
{
    return k16_guard(ctx, [&]() -> int {
    OneShotGuard one_shot(ctx);
    if (!ctx || group != K16_G1 || !d_table || !d_scalars) return K16_ERR_ARG;
    const unsigned c = fixed_base_c(n);
    if (!c) {
        ctx->err = "fixed-base msm: unsupported n";
        return K16_ERR_ARG;
    }
    K16_HIP(ctx, hipSetDevice(ctx->device));
    int idx;
    {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        if (ctx->pend_count == k16_ctx::PEND_SLOTS) {
            ctx->err = "k16_msm_enqueue: too many MSMs in flight; call k16_msm_finish";
            return K16_ERR_ARG;
        }
        idx = (ctx->pend_head + ctx->pend_count) % k16_ctx::PEND_SLOTS;
    }
    k16_ctx::Pend pd;
    pd.group      = group;
    pd.n          = n;
    pd.slot       = idx;
    pd.c          = c;
    pd.flat       = true;
    ctx->enq_slot = idx;
    int rc        = k16_msm_enqueue_fixed_g1(ctx, d_table, d_scalars, n, c);
    if (rc) return rc;
    pd.w     = ctx->pend_wr;
    pd.nbits = ctx->pend_nbits;
    pd.mlog  = ctx->pend_mlog;
    K16_HIP(ctx, hipEventRecord(ctx->pend_ev[idx], k16_lane_stream(ctx, ctx->cur_lane)));
    {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        ctx->pend[idx] = pd;
        ctx->pend_count++;
    }
    return K16_OK;
    });
}

extern "C" int k16_msm_set_lane(k16_ctx* ctx, int lane)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || lane < 0 || lane >= k16_ctx::N_LANES) return K16_ERR_ARG;
    ctx->cur_lane = lane;
    return K16_OK;
    });
}

// The next k16_msm_enqueue* leaves the points whose mask bit is set out of its bucket sort (k16_msm_zero_row_mask of the
// table: (0,0) rows add nothing, curve.cpp:185-250, but as sorted entries they cost a lane of every addition they sit
// beside).  One call covers one enqueue.  n <= 2^24, not for fixed-base tables.
extern "C" int k16_msm_set_zero_row_mask(k16_ctx* ctx, const void* d_mask)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx) return K16_ERR_ARG;
    ctx->skip_next = (const uint64_t*)d_mask;
    return K16_OK;
    });
}

// The next k16_msm_enqueue_prepared reads the bucket sort another lane has made of the SAME scalar array (same device
// pointer, n, window size) instead of sorting again -- several tables indexed by the same scalars, e.g. the A / B1 / B2 / C
// tables of a Groth16 key (groth16.cpp:88-112 runs four MSMs over one witness).
//   derive = 0  the lists as they are; the zero-row mask named for this enqueue must be the one that sort was made with
//   derive = 1  bucket lists of this lane's own, built from that lane's PARTITION without the rows in this enqueue's
//               zero-row mask (k16_msm_set_zero_row_mask; a superset of the owner's mask): nothing is partitioned again, and
//               the table's accumulation walks live rows only.  Other lanes may then share THIS lane's lists with derive = 0.
extern "C" int k16_msm_sort_from_lane(k16_ctx* ctx, int lane, int derive)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || lane < 0 || lane >= k16_ctx::N_LANES) return K16_ERR_ARG;
    if (derive) {
        ctx->derive_lane = lane;
    } else {
        ctx->reuse_sort      = true;
        ctx->reuse_sort_lane = lane;
    }
    return K16_OK;
    });
}

extern "C" int k16_msm_set_window_bits(k16_ctx* ctx, unsigned c)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx) return K16_ERR_ARG;
    if (c != 0 && (c < MIN_C || c > MAX_C)) return K16_ERR_ARG;
    ctx->forced_c = c;
    return K16_OK;
    });
}

namespace {
// host-side cost of the MSM entry points (names "host_enqueue", "host_finish_wait", "host_finish_combine" in
// k16_kernel_stats_get), so that a benchmark can tell a launch-bound host from a busy GPU
struct HostTimer {
    k16_ctx*                              c;
    const char*                           name;
    std::chrono::steady_clock::time_point t0;
    HostTimer(k16_ctx* cx, const char* nm) : c(cx), name(nm), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer()
    {
        if (!c->stats_on) return;
        std::lock_guard<std::mutex> lk(c->ring_mu);
        auto&        st = c->stats[name];
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        st.launches++;
        st.total_ms += ms;
        auto& mx = c->stats[std::string(name) + "_max"]; // the slowest call since the last reset (total_ms holds it)
        mx.launches = 1;
        if (ms > mx.total_ms) mx.total_ms = ms;
    }
};
} // namespace

static int msm_enqueue_any(k16_ctx* ctx, int group, const void* d_bases, const void* d_scalars, uint64_t n, int prepared)
{
    OneShotGuard one_shot(ctx);
    if (!ctx || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
    HostTimer ht(ctx, "host_enqueue");
    K16_HIP(ctx, hipSetDevice(ctx->device)); // the calling thread may be new (bench.py enqueues from a second thread)
    if (n >= (1ull << 32) / 80) { // index / offset arithmetic is 32-bit: n * W must stay below 2^32
        ctx->err = "k16_msm: n too large for one device call; shard it";
        return K16_ERR_ARG;
    }
    int idx;
    {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        if (ctx->pend_count == k16_ctx::PEND_SLOTS) {
            ctx->err = "k16_msm_enqueue: too many MSMs in flight; call k16_msm_finish";
            return K16_ERR_ARG;
        }
        idx = (ctx->pend_head + ctx->pend_count) % k16_ctx::PEND_SLOTS; // = (number enqueued so far) mod slots
    }
    k16_ctx::Pend pd;
    pd.group = group;
    pd.n     = n;
    pd.slot  = idx;
    int rc   = K16_OK;
    if (n != 0) {
        unsigned c    = choose_c(ctx, n);
        pd.c          = c;
        pd.w          = n_windows(c);
        ctx->enq_slot = idx;
        rc = group == K16_G1 ? k16_msm_enqueue_g1(ctx, d_bases, d_scalars, n, c, prepared)
                             : k16_msm_enqueue_g2(ctx, d_bases, d_scalars, n, c, prepared);
        if (rc) return rc;
        pd.nbits = ctx->pend_nbits;
        pd.mlog  = ctx->pend_mlog;
        K16_HIP(ctx, hipEventRecord(ctx->pend_ev[idx], k16_lane_stream(ctx, ctx->cur_lane)));
    }
    {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        ctx->pend[idx] = pd;
        ctx->pend_count++;
    }
    return K16_OK;
}

extern "C" int k16_msm_enqueue(k16_ctx* ctx, int group, const void* d_bases, const void* d_scalars, uint64_t n)
{
    return k16_guard(ctx, [&]() -> int {
    return msm_enqueue_any(ctx, group, d_bases, d_scalars, n, 0);
    });
}
extern "C" int k16_msm_enqueue_prepared(k16_ctx* ctx, int group, const void* d_prepared, const void* d_scalars,
                                        uint64_t n)
{
    return k16_guard(ctx, [&]() -> int {
    return msm_enqueue_any(ctx, group, d_prepared, d_scalars, n, 1);
    });
}
// Scalar-class MSM of one prepared table (msm_kernels.inc "Scalar-class MSM", msm_classes.hip): masked sums of the wires
// below 256 + the ordinary MSM over the compacted wide scalars; results through k16_msm_finish like any other MSM.
// phase 0: all of it.  The prover splits it: phase 1 (masked sums only; a staging slot is reserved) for all its tables first,
// then phase 2 (wide part, the MSM joins the result queue) in the same order.
int k16_msm_classified_phase(k16_ctx* ctx, int group, const void* d_prepared, const k16_scalar_classes* cls, int set, int phase)
{
    OneShotGuard one_shot(ctx);
    if (!ctx || !cls || cls->ctx != ctx || (group != K16_G1 && group != K16_G2) || set < 0 || set >= cls->n_sets ||
        (cls->n && !d_prepared))
        return K16_ERR_ARG;
    HostTimer ht(ctx, "host_enqueue");
    K16_HIP(ctx, hipSetDevice(ctx->device));
    int idx;
    {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        const int ahead = phase == 1 ? ctx->pend_reserved : 0; // phase 2 completes the OLDEST reservation: the next ring position
        if (ctx->pend_count + ahead >= k16_ctx::PEND_SLOTS || (phase == 2 && ctx->pend_reserved == 0)) {
            ctx->err = "k16_msm_enqueue: too many MSMs in flight; call k16_msm_finish";
            return K16_ERR_ARG;
        }
        idx = (ctx->pend_head + ctx->pend_count + ahead) % k16_ctx::PEND_SLOTS;
        if (phase == 1) ctx->pend_reserved++;
    }
    k16_ctx::Pend pd;
    pd.group = group;
    pd.n     = cls->n;
    pd.slot  = idx;
    if (cls->n != 0) {
        // window size of the wide part (a Keyless witness has ~27 k wide values): log2(n) - 1, i.e. 13 there -- few entries
        // per bucket, so a lane's chain of additions is short (these MSMs are latency-bound beside the polynomial chain)
        // and no bucket spans segments; c = 11 would be 10 % less work and twice the chain
        unsigned c = ctx->forced_c;
        if (c < MIN_C || c > MAX_C) {
            unsigned lg = 0;
            while ((std::max<uint64_t>(cls->n_wide, 1) >> (lg + 1)) != 0) lg++;
            c = (unsigned)std::min<int>(MAX_C, std::max<int>(MIN_C, (int)lg - 1));
        }
        ctx->enq_slot    = idx;
        bool has_wide    = false;
        int  rc = group == K16_G1 ? k16_msm_enqueue_classified_g1(ctx, d_prepared, cls, set, c, &has_wide, phase)
                                  : k16_msm_enqueue_classified_g2(ctx, d_prepared, cls, set, c, &has_wide, phase);
        if (rc) return rc;
        pd.cls    = cls;
        pd.narrow = k16_scalar_classes::BITS;
        if (has_wide) {
            pd.c     = c;
            pd.w     = n_windows(c);
            pd.nbits = ctx->pend_nbits;
            pd.mlog  = ctx->pend_mlog;
        }
        if (phase != 1) K16_HIP(ctx, hipEventRecord(ctx->pend_ev[idx], k16_lane_stream(ctx, ctx->cur_lane)));
    }
    if (phase != 1) {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        ctx->pend[idx] = pd;
        ctx->pend_count++;
        if (phase == 2) ctx->pend_reserved--;
    }
    return K16_OK;
}
extern "C" int k16_msm_enqueue_classified(k16_ctx* ctx, int group, const void* d_prepared, const k16_scalar_classes* cls, int set)
{
    return k16_guard(ctx, [&]() -> int {
    if (ctx && ctx->pend_reserved) {
        ctx->err = "k16_msm_enqueue_classified: a split classified MSM is still open";
        return K16_ERR_ARG;
    }
    return k16_msm_classified_phase(ctx, group, d_prepared, cls, set, 0);
    });
}

extern "C" int k16_msm_bases_prepare(k16_ctx* ctx, int group, const void* d_bases, uint64_t n, void* d_out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || (group != K16_G1 && group != K16_G2) || (n && (!d_bases || !d_out))) return K16_ERR_ARG;
    if (group == K16_G1) return k16_msm_prepare_g1(ctx, d_bases, n, d_out, nullptr);
    return k16_msm_prepare_g2(ctx, d_bases, n, d_out, nullptr);
    });
}

// The head entry stays in the ring (pend_count unchanged) until its staged partial sums have been copied out of the
// pinned slot: an enqueuing thread computes its slot as (head + count) % PEND_SLOTS, so popping first would let it
// re-record the event and queue a download into the slot this thread is still waiting on / reading.
// expect_group >= 0: fail (without consuming anything) when the head MSM is of the other group -- a G2 result is 256
// bytes and must never be written into a caller's 128-byte G1 buffer.
static int msm_finish_any(k16_ctx* ctx, int expect_group, void* h_out_xyzz, void* h_out_affine)
{
    if (!ctx) return K16_ERR_ARG;
    k16_ctx::Pend pd;
    {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        if (ctx->pend_count == 0) return K16_ERR_ARG;
        pd = ctx->pend[ctx->pend_head];
        if (expect_group >= 0 && pd.group != expect_group) {
            ctx->err = "k16_msm_finish: the oldest MSM in flight belongs to the other group";
            return K16_ERR_ARG;
        }
    }
    auto pop = [&]() {
        std::lock_guard<std::mutex> lk(ctx->ring_mu);
        ctx->pend_head = (ctx->pend_head + 1) % k16_ctx::PEND_SLOTS;
        ctx->pend_count--;
    };
    const int group = pd.group;
    if (pd.n == 0) {
        pop();
        if (group == K16_G1) {
            G1Xyzz z = G1Xyzz::zero();
            if (h_out_xyzz) memcpy(h_out_xyzz, &z, sizeof z);
            if (h_out_affine) memset(h_out_affine, 0, sizeof(G1Aff));
        } else {
            G2Xyzz z = G2Xyzz::zero();
            if (h_out_xyzz) memcpy(h_out_xyzz, &z, sizeof z);
            if (h_out_affine) memset(h_out_affine, 0, sizeof(G2Aff));
        }
        return K16_OK;
    }
    {
        HostTimer hw(ctx, "host_finish_wait");
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = k16_event_wait(ctx, ctx->pend_ev[pd.slot]); // only this MSM's results; later ones keep running
        if (e != hipSuccess) {
            pop(); // the entry is consumed either way: a failed MSM must not be handed to the next caller
            ctx->err = std::string("k16_msm_finish: ") + hipGetErrorString(e);
            return K16_ERR_HIP;
        }
    }
    HostTimer hc(ctx, "host_finish_combine");
    const char*    src = (const char*)ctx->pinned + (size_t)pd.slot * k16_ctx::SLOT_BYTES;
    const unsigned cnt = pd.w * (pd.nbits + 2);
    if (pd.cls_overflow || (pd.cls && pd.cls->h_flags[0])) { // the classification met more wide scalars than its caller announced: some were dropped
        pop();
        ctx->err = "k16_msm_finish: the scalar classes were built with a wide-scalar bound below the actual count";
        return K16_ERR_ARG;
    }
    // scalar-class MSM: + sum_b 2^b S_b over the masked sums staged behind the window sums (Horner from the top bit)
    auto add_narrow = [&](auto& r, const auto& S) {
        auto acc = S[pd.narrow - 1];
        for (int b = (int)pd.narrow - 2; b >= 0; b--) acc = padd(pdbl(acc), S[b]);
        r = padd(r, acc);
    };
    if (group == K16_G1) {
        // the G1 kernels work in the radix-2^29 / R' domain: bring the few window/bit sums back to the
        // reference's canonical Montgomery form first (exact conversion)
        std::vector<Xyzz9> raw(cnt);
        memcpy(raw.data(), src, (size_t)cnt * sizeof(Xyzz9));
        Xyzz9 nraw[k16_scalar_classes::BITS];
        if (pd.narrow) memcpy(nraw, src + k16_ctx::NARROW_OFF, pd.narrow * sizeof(Xyzz9));
        pop(); // the slot may be reused from here on
        std::vector<G1Xyzz> T(cnt);
        const std::function<void(unsigned)> prep = [&](unsigned w) {
            for (unsigned i = w * (pd.nbits + 2); i < (w + 1) * (pd.nbits + 2); i++) T[i] = xyzz9_to_canonical(raw[i]);
        };
        g_window_prep = &prep;
        struct Clear {
            ~Clear() { g_window_prep = nullptr; }
        } clear;
        G1Xyzz r;
        if (pd.flat)
            flat_combine_host<Fq>(ctx, T.data(), pd.w, pd.c, pd.nbits, pd.mlog, &r);
        else
            horner_host<Fq>(ctx, T.data(), pd.w, pd.c, pd.nbits, pd.mlog, &r);
        if (pd.narrow) {
            G1Xyzz S[k16_scalar_classes::BITS];
            for (unsigned b = 0; b < pd.narrow; b++) S[b] = xyzz9_to_canonical(nraw[b]);
            add_narrow(r, S);
        }
        if (h_out_xyzz) memcpy(h_out_xyzz, &r, sizeof r);
        if (h_out_affine) {
            G1Aff a = to_affine(r);
            memcpy(h_out_affine, &a, sizeof a);
        }
    } else {
        std::vector<Xyzz<Fq2n>> raw(cnt);
        memcpy(raw.data(), src, (size_t)cnt * sizeof(Xyzz<Fq2n>));
        Xyzz<Fq2n> nraw[k16_scalar_classes::BITS];
        if (pd.narrow) memcpy((void*)nraw, src + k16_ctx::NARROW_OFF, pd.narrow * sizeof(Xyzz<Fq2n>));
        pop();
        std::vector<G2Xyzz> T(cnt);
        auto canon2 = [](const Xyzz<Fq2n>& p9) {
            return p9.is_zero() ? G2Xyzz::zero()
                                : G2Xyzz{fq2n_to_canonical(p9.x), fq2n_to_canonical(p9.y), fq2n_to_canonical(p9.zz),
                                         fq2n_to_canonical(p9.zzz)};
        };
        const std::function<void(unsigned)> prep = [&](unsigned w) {
            for (unsigned i = w * (pd.nbits + 2); i < (w + 1) * (pd.nbits + 2); i++) T[i] = canon2(raw[i]);
        };
        g_window_prep = &prep;
        struct Clear {
            ~Clear() { g_window_prep = nullptr; }
        } clear;
        G2Xyzz r;
        horner_host<Fq2>(ctx, T.data(), pd.w, pd.c, pd.nbits, pd.mlog, &r);
        if (pd.narrow) {
            G2Xyzz S[k16_scalar_classes::BITS];
            for (unsigned b = 0; b < pd.narrow; b++) S[b] = canon2(nraw[b]);
            add_narrow(r, S);
        }
        if (h_out_xyzz) memcpy(h_out_xyzz, &r, sizeof r);
        if (h_out_affine) {
            G2Aff a = to_affine(r);
            memcpy(h_out_affine, &a, sizeof a);
        }
    }
    return K16_OK;
}

extern "C" int k16_msm_finish(k16_ctx* ctx, void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(ctx, [&]() -> int {
    return msm_finish_any(ctx, -1, h_out_xyzz, h_out_affine);
    });
}
extern "C" int k16_msm_finish_group(k16_ctx* ctx, int group, void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(ctx, [&]() -> int {
    if (group != K16_G1 && group != K16_G2) return K16_ERR_ARG;
    return msm_finish_any(ctx, group, h_out_xyzz, h_out_affine);
    });
}
extern "C" int k16_msm_pending(k16_ctx* ctx)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx) return K16_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->ring_mu);
    return ctx->pend_count;
    });
}
// Error recovery: wait for and drop every MSM still in flight, forget any bucket sort marked for reuse and go back to
// lane 0, so that the next caller starts from a clean queue (a prover that failed half-way must not leave its MSMs behind).
extern "C" int k16_msm_abort_all(k16_ctx* ctx)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx) return K16_ERR_ARG;
    int rc = K16_OK;
    while (k16_msm_pending(ctx) > 0) {
        int r = msm_finish_any(ctx, -1, nullptr, nullptr);
        if (r && !rc) rc = r;
    }
    ctx->reuse_sort      = false;
    ctx->reuse_sort_lane = -1;
    ctx->cur_lane        = 0;
    ctx->forced_seg      = 0;
    ctx->pend_reserved   = 0;
    ctx->remap_next      = nullptr;
    ctx->skip_next       = nullptr;
    ctx->acc_skip_next   = nullptr;
    ctx->derive_lane     = -1;
    ctx->hs_next[0] = ctx->hs_next[1] = ctx->hs_next[2] = nullptr;
    (void)hipSetDevice(ctx->device);
    for (auto& L : ctx->lanes)
        if (L.stream) (void)hipStreamSynchronize(L.stream);
    return rc;
    });
}

// One device call handles up to 2^24 points on the fast (LDS partition) sort.  A larger MSM on ONE GPU is the same
// sharding as across GPUs, done in time: contiguous chunks, alternating lanes so that a chunk's sort
// overlaps its predecessor's accumulation, and an EC-add fold of the per-chunk results.
constexpr uint64_t MSM_CHUNK = 1ull << 24;

extern "C" int k16_msm(k16_ctx* ctx, int group, const void* d_bases, const void* d_scalars, uint64_t n,
                       void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
    if (n <= MSM_CHUNK) {
        int rc = k16_msm_enqueue(ctx, group, d_bases, d_scalars, n);
        if (rc) return rc;
        return k16_msm_finish(ctx, h_out_xyzz, h_out_affine);
    }
    if (k16_msm_pending(ctx) != 0) {
        ctx->err = "k16_msm: a chunked (n > 2^24) call needs an empty MSM queue";
        return K16_ERR_ARG;
    }
    const size_t   pb       = group == K16_G1 ? sizeof(G1Aff) : sizeof(G2Aff);
    const size_t   xb       = group == K16_G1 ? sizeof(G1Xyzz) : sizeof(G2Xyzz);
    const uint64_t chunks   = (n + MSM_CHUNK - 1) / MSM_CHUNK;
    const int      saved    = ctx->cur_lane;
    std::vector<unsigned char> parts((size_t)chunks * xb);
    int      rc = K16_OK;
    uint64_t enq = 0, fin = 0;
    auto enqueue_next = [&]() -> int {
        const uint64_t lo = enq * MSM_CHUNK, cnt = std::min<uint64_t>(MSM_CHUNK, n - lo);
        ctx->cur_lane = (int)(enq % 2); // two lanes: two chunks in flight
        enq++;
        return k16_msm_enqueue(ctx, group, (const char*)d_bases + lo * pb, (const char*)d_scalars + lo * 32, cnt);
    };
    rc = enqueue_next();
    while (!rc && fin < chunks) {
        if (enq < chunks) rc = enqueue_next();
        if (!rc) rc = k16_msm_finish(ctx, parts.data() + (size_t)fin * xb, nullptr);
        fin++;
    }
    ctx->cur_lane = saved;
    if (rc) {
        (void)k16_msm_abort_all(ctx); // drain
        return rc;
    }
    return k16_points_sum(group, parts.data(), chunks, h_out_xyzz, h_out_affine);
    });
}

extern "C" int k16_msm_host(k16_ctx* ctx, int group, const void* h_bases, const void* h_scalars, uint64_t n,
                            void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
    if (n == 0) return k16_msm(ctx, group, nullptr, nullptr, 0, h_out_xyzz, h_out_affine);
    size_t pb = (size_t)n * (group == K16_G1 ? sizeof(G1Aff) : sizeof(G2Aff));
    void * db = nullptr, *ds = nullptr;
    K16_HIP(ctx, hipMalloc(&db, pb));
    hipError_t e = hipMalloc(&ds, (size_t)n * 32);
    if (e != hipSuccess) {
        (void)hipFree(db);
        ctx->err = "hipMalloc scalars";
        return K16_ERR_HIP;
    }
    int rc = K16_OK;
    if (hipMemcpyAsync(db, h_bases, pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(ds, h_scalars, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        ctx->err = "hipMemcpyAsync h2d";
        rc       = K16_ERR_HIP;
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = K16_ERR_HIP; // uploads ran on lane 0's stream
    if (!rc) rc = k16_msm(ctx, group, db, ds, n, h_out_xyzz, h_out_affine);
    (void)hipDeviceSynchronize();
    (void)hipFree(db);
    (void)hipFree(ds);
    return rc;
    });
}

extern "C" int k16_points_sum(int group, const void* h_parts, uint64_t count, void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(nullptr, [&]() -> int {
    if (group == K16_G1) {
        G1Xyzz acc = G1Xyzz::zero();
        for (uint64_t i = 0; i < count; i++) {
            G1Xyzz p;
            memcpy(&p, (const char*)h_parts + i * sizeof p, sizeof p);
            acc = padd(acc, p);
        }
        if (h_out_xyzz) memcpy(h_out_xyzz, &acc, sizeof acc);
        if (h_out_affine) {
            G1Aff a = to_affine(acc);
            memcpy(h_out_affine, &a, sizeof a);
        }
        return K16_OK;
    }
    if (group == K16_G2) {
        G2Xyzz acc = G2Xyzz::zero();
        for (uint64_t i = 0; i < count; i++) {
            G2Xyzz p;
            memcpy(&p, (const char*)h_parts + i * sizeof p, sizeof p);
            acc = padd(acc, p);
        }
        if (h_out_xyzz) memcpy(h_out_xyzz, &acc, sizeof acc);
        if (h_out_affine) {
            G2Aff a = to_affine(acc);
            memcpy(h_out_affine, &a, sizeof a);
        }
        return K16_OK;
    }
    return K16_ERR_ARG;
    });
}
