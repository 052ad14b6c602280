// ctx.h -- internal context shared by the HIP translation units of libk16.so
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <new>
#include <exception>
#include <tuple>
#include <string>
#include <vector>
#include <stdint.h>
#include "../../include/k16.h"
#include "bn254_curve.h"
#include "bn254_fq9.h"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>

#include "host_pool.h"

struct k16_devbuf {
    void*  p     = nullptr;
    size_t bytes = 0;
};

struct k16_kstat {
    uint64_t launches = 0;
    double   total_ms = 0;
};

struct k16_ntt_table {
    uint32_t  s      = 0; // log2 size
    k16::Fr*  roots  = nullptr; // device, 2^s entries, canonical Montgomery (R = 2^256)
    uint32_t* roots9 = nullptr; // device, the same roots as R' values (x * 2^261 mod r, < 2r) in nine 29-bit limbs each (36 bytes), see ntt.hip
    // Round 6: the twiddles of the LATE stages once more, stage-major and contiguous (ntt.hip k_build_stage9): stage s' >= 17 of a
    // transform uses root(s', j) = roots[j << (s - s')] for j < 2^(s'-1) -- consecutive butterflies read entries 2^(s - s')
    // apart, a 36-byte gather with a stride of 72 B ... 1.1 KB over a 151 MB table (4 M loads per 2^21 transform, as many bytes
    // as the data itself).  stage9 holds root(s', j) at entry 2^(s'-1) - 2^16 + j for 17 <= s' < s: 32 consecutive butterflies
    // read 1152 contiguous bytes.  Null for tables below 2^18 or with K16_NTT_NO_STAGE_TABLES.
    uint32_t* stage9 = nullptr;
    k16::Fr   pow2inv[34];
    k16::Fq9  pow2inv9[34];     // 2^-k as Fr9
};

// Every environment switch of the library, read ONCE when a context is created (k16_ctx_create) and
// kept in the context: no getenv on any per-MSM, per-transform or per-proof path (getenv races with a host program's setenv
// and costs a lock per call), and an object made from a context (prover, verifying key) sees the values its context was made
// with.  All of these select between code paths that give the SAME results (alternative / reference implementations kept
// parity-tested, rejected scheduling experiments kept so that they can be re-measured) or switch diagnostics on.  Switches
// that change a RESULT on purpose (measurement probes, fault injection) do not exist in libk16.so: they are compiled only
// into the lab / testing builds (-DK16_LAB, -DK16_TESTING).
struct k16_tuning {
    bool     atomic_sort = false, no_fused_convert = false, no_staged_sort = false, fused_bins = false, x8 = false;
    bool     no_l1_prefetch = false, ntt_tail_small = false, ntt_unfused = false, ntt_no_stage_tables = false;
    bool     no_fixed_base = false, no_stream_priority = false, b_sort = false, b_derive = false, no_skip_zero_rows = false;
    bool     classes = false, no_warmup = false, spmv_full = false, fused_hscalars = false, no_split_classes = false;
    bool     b2_first = false, no_acc_skip = false;
    // round 6 scheduling experiments (identical results): lane of the H MSM (default 1, behind C's MSM), lane of
    // the B1 MSM (default 0, behind A's), the H MSM's wait for the chain issued behind its sort's memset instead of in front
    int      g2_acc_split = 1; // lane pairs per segment in the witness MSMs' G2 accumulation (1, 2 or 4; msm_kernels.inc k_accumulate_split)
    int      h_lane = 1, b1_lane = 0, witness_seg = 0; // witness_seg: segment length of the witness MSMs' accumulations (default 32)
    bool     h_wait_first = false;
    bool     trace = false, trace_enq = false, trace_host = false, verify_no_coop = false, verify_coop_trace = false;
    int      seg = 0, wsum_mlog = -1, witness_c = 0, ntt_tile_log = 0, narrow_chain = 0, narrow_chain_g2 = 0;
    uint64_t verify_coop_max = 2048;
    static k16_tuning from_env();
};

struct k16_ctx {
    int         device = 0;
    k16_tuning  tune;
    hipStream_t stream = nullptr;
    hipEvent_t  ev_a = nullptr, ev_b = nullptr;     // k16_timer_*
    // kernel stats: event pairs are recorded without synchronising; resolved in k16_kernel_stats_get
    std::vector<hipEvent_t> ks_pool;
    std::vector<std::pair<std::string, size_t>> ks_pending; // (name, index of the start event in ks_pool)
    size_t      ks_used = 0;
    int         stats_on = 0; // 0 off, 1 every named stage, 2 only "msm_accumulate" (two HIP events per MSM instead of eight)
    std::map<std::string, k16_kstat> stats;
    std::string err;
    unsigned    forced_c = 0;

    // MSM workspace (grown on demand, reused across calls)
    // MSM lanes: independent (stream, workspace) pairs.  MSMs enqueued on different lanes may overlap on the
    // GPU -- the fold / weighted-sum stages are latency-bound chains on few lanes and leave most CUs idle, so a
    // second MSM's sort or accumulation fills them.  Lane 0's stream is also ctx->stream.
    static constexpr int N_LANES = 4;
    struct Lane {
        hipStream_t stream = nullptr;
        k16_devbuf  ws_counts, ws_offsets, ws_cursor, ws_sorted, ws_segoff, ws_segbucket, ws_partial, ws_big, ws_misc,
            ws_lvl_a, ws_lvl_b, ws_lvl_c, ws_lvl_d, ws_scan, ws_conv, ws_narrow;
        // bucket sort still valid in this lane's workspace (same scalars, n, c): see reuse_sort
        // hipGraph replay of the launch-bound parts of an MSM (see graphs_on): key -> state / executable graph
        struct GraphEntry {
            int            state = 0; // 0 unseen, 1 ran eagerly once (workspace sized), 2 captured, -1 capture failed: eager
            hipGraphExec_t exec  = nullptr;
        };
        typedef std::tuple<int, const void*, uint64_t, unsigned, unsigned, unsigned, int> GraphKey;
        std::map<GraphKey, GraphEntry> graphs;
        uint64_t                       graphs_gen = 0; // workspace generation the cached graphs were captured for
        hipEvent_t  sort_done = nullptr; // recorded on `stream` after every bucket sort (cross-lane reuse waits on it)
        hipEvent_t  acc_done  = nullptr; // recorded after every bucket accumulation (see serialize_acc)
        hipEvent_t  lvl1_done = nullptr; // ... after the first level of the weighted bucket sum, and after the whole
        hipEvent_t  tail_done = nullptr; //     reduction (see acc_fence_mode)
        const void* sorted_scalars = nullptr;
        uint64_t    sorted_n = 0;
        unsigned    sorted_c = 0;
        const uint64_t* sorted_skip = nullptr; // zero-row mask that sort was made with
        bool        sorted_plain_partition = false; // ... and its partition (ws_lvl_d) can be read by a derived sort
    };
    Lane lanes[N_LANES];
    int  cur_lane = 0; // lane of the next k16_msm_enqueue*
    std::vector<hipStream_t> placeholder_streams; // k16_ctx_create_ex: created before the context's own, destroyed with it
    bool yielding_waits = false; // K16_OPT_YIELDING_WAITS: host waits poll + sleep instead of spinning inside the runtime (k16_event_wait)
    hipEvent_t wait_after_memset = nullptr; // one-shot: the next bucket sort waits for this event BEHIND its tables' memset
    void* pinned = nullptr;     // small pinned host staging buffer (coherent, mapped into the device's address space)
    void* pinned_dev = nullptr; // its device-side address: the last kernel of an MSM writes its <= 240 partial sums straight
                                // into the staging slot (a hipMemcpyAsync D2H was observed to BLOCK the enqueuing thread for
                                // 7-13 ms a few times per process while the runtime set up its copy path)
    size_t pinned_bytes = 0;

    // state of the MSM currently enqueued (k16_msm_enqueue -> k16_msm_finish)
    // MSMs enqueued and not yet finished (FIFO): several can be in flight on the stream, each with its own
    // slot of the pinned staging buffer and an event recorded after its device-to-host copy, so the host
    // tail of one MSM (conversion + Horner) overlaps the next MSM's kernels
    static constexpr int    PEND_SLOTS = 8;
    static constexpr size_t SLOT_BYTES = 128 * 1024;
    struct Pend {
        int      group = -1;
        unsigned c = 0, w = 0, nbits = 0, mlog = 0;
        bool     flat = false; // fixed-base MSM: w = pseudo-windows of the single bucket set
        uint64_t n = 0;
        int      slot = 0;
        // scalar-class MSM (k16_msm_enqueue_classified): the staging slot also holds `narrow` masked point sums S_b (the wires
        // below 2^narrow with bit b set) at NARROW_OFF; the result is the windows' value + sum_b 2^b S_b
        unsigned narrow = 0;
        const struct k16_scalar_classes* cls = nullptr;
        bool     cls_overflow = false; // the classes object went away before the finish: its overflow flag, copied (k16_scalar_classes_destroy)
    };
    static constexpr size_t NARROW_OFF = SLOT_BYTES - 4096; // 8 points of <= 288 bytes at the end of a staging slot
    // One thread may enqueue while another finishes (bench.py does: launches on a slow host then overlap the wait for the
    // GPU): the ring bookkeeping and the host timers are guarded; everything else an enqueue touches is its own.
    std::mutex ring_mu;
    std::mutex verify_mu; // the latency path of k16_verify_batch: pinned staging area + the key's small device buffers
    Pend       pend[PEND_SLOTS];
    int        pend_head = 0, pend_count = 0; // ring: oldest at pend_head
    int        pend_reserved = 0;             // classified MSMs whose narrow part is enqueued and whose wide part is still to come (k16_msm_classified_phase)
    hipEvent_t pend_ev[PEND_SLOTS] = {};
    int        enq_slot = 0;                  // staging slot of the MSM being enqueued
    unsigned   pend_mlog = 0;
    unsigned   pend_nbits = 0;                // written by the kernels' host code for the MSM being enqueued
    unsigned   pend_wr = 0;                   // (pseudo-)windows whose partial sums were staged

    // bucket sort of the previous MSM still valid in the workspace (same scalars, n, c): the prover's A / B1 / B2
    // MSMs all use the witness as scalars, so the sort is done once (set by the prover, cleared by every sort)
    bool        reuse_sort = false;
    unsigned    forced_seg = 0; // accumulate segment length override (0: automatic)
    // K16_OPT_GRAPHS: an MSM is ~55 host calls (launches, memsets, events); on a host whose launches are slow (a busy
    // node: 25 us per call measured, against 2 us) that alone is 1.4 ms per MSM.  With graphs on, the sort and the
    // fold + reduction + download sequences are captured once per (lane, shape, staging slot) and replayed with one
    // hipGraphLaunch each; the accumulation stays an ordinary launch (its HIP-event timing keeps working).
    bool        graphs_on = false;
    bool        capturing = false; // inside hipStreamBeginCapture: no event-based statistics, no allocation
    uint64_t    ws_gen = 0;        // bumped by every workspace reallocation: cached graphs hold the old pointers
    // lane whose sort the next MSM reuses (-1: the MSM's own lane).  With another lane the MSM reads that lane's index
    // lists but runs on its own stream with its own partial / reduction buffers, i.e. concurrently with that lane's MSMs.
    int         reuse_sort_lane = -1;
    // table-row indirection of the NEXT bucket sort (consumed by it): the sort's scalars are a compacted subset, point k of
    // it is row remap_next[k] of the tables (k16_msm_enqueue_classified: the wide scalars of a witness)
    const uint32_t* remap_next = nullptr;
    // zero-row mask of the NEXT bucket sort (consumed by it; an MSM that REUSES a sort must name the mask it was made with)
    const uint64_t* skip_next = nullptr;
    // zero-row mask of the NEXT MSM's own table, applied in its accumulation (for a table that reuses another table's sort)
    const uint64_t* acc_skip_next = nullptr;
    int             derive_lane   = -1; // next enqueue: bucket lists of its own from that lane's partition, minus skip_next's rows
    // workgroups of a 1024-element NTT pass per CU (ntt.hip): 4 is what fits; 3 (K16_OPT_SHARED_GPU) leaves registers for a wave of
    // another prover's bucket accumulation beside them
    unsigned        ntt_wg_per_cu = 4, ntt_wg_per_cu_default = 4;
    // next enqueue: its scalars do not exist yet -- scalar i = fromMontgomery(hs_next[0][i] * hs_next[1][i] - hs_next[2][i]) over Fr
    // (packed R' values), formed by the sort's counting pass and written to d_scalars (the prover's H MSM, groth16.cpp:266-283)
    const void*     hs_next[3]    = {nullptr, nullptr, nullptr};
    // K16_SERIALIZE_ACC=1 (bench.py sets it): a lane's bucket accumulation waits for the previous lane's.  Two of these
    // chip-filling kernels never overlap anyway (kernel traces: the second starts when the first ends), so nothing is
    // lost, but the HIP events that time the kernel on its own stream then bracket its execution only -- without the fence
    // the interval also contains the time the launch sits behind the other lane's accumulation.
    bool        serialize_acc = false;
    // slots per lane of the first weighted-sum level are capped at 2^wsum_mlog_cap: 8 (3) gives the shortest single MSM;
    // 16 (4) does 12 % less reduction work with chains twice as long -- +5 % for pipelined MSMs, +4 % latency for one
    unsigned    wsum_mlog_cap = 3;
    hipEvent_t  last_acc_done = nullptr;
    // what a lane's accumulation waits for when serialize_acc is on: 0 the previous MSM's accumulation, 1 its first
    // weighted-sum level (the previous tail's fold + level 1 run alone, its bit sums beside this accumulation), 2 its whole
    // reduction.  A chip-filling accumulation starves every kernel of the other lanes that starts beside it (kernel traces:
    // a fold launched next to an accumulation ends when the accumulation ends), so with mode 0 the tails of ALL queued MSMs
    // wait for ALL queued accumulations.
    int         acc_fence_mode = 0;
    hipEvent_t  last_lvl1_done = nullptr, last_tail_done = nullptr;

    // Unused dynamic LDS requested for the bucket accumulation, to cap ITS occupancy (K16_ACC_LDS, bytes per 128-thread
    // workgroup: 36864 -> 4 workgroups = 2 waves/SIMD per CU instead of the 3 its 159 VGPRs allow).  The registers a
    // third wave would take stay free for the other lanes' sort / fold / reduction kernels, which otherwise wait for a
    // whole accumulate workgroup to retire before one of their waves fits.
    unsigned    acc_lds_bytes = 0;
    unsigned    acc_grid_cap  = 0; // K16_ACC_GRID: at most this many (persistent, grid-stride) accumulate workgroups
    // every kernel of the bucket sort in <= 32 VGPRs, so that it is resident BESIDE another lane's bucket accumulation
    // (msm_kernels.inc, "lean sort"); set with K16_OPT_PIPELINED_MSM, or K16_LEAN_SORT=0/1
    bool        lean_sort     = false;
    bool        wc_sort       = false; // K16_WC_SORT: write-combining scatter pass of the partition (k_part_wc)
    unsigned    acc_dyn_grid  = 0; // K16_ACC_DYN: persistent accumulate grid of this many workgroups with dynamic chunk fetch

    std::map<uint32_t, k16_ntt_table> ntt_tables;
    // the per-window combine of an MSM's partial sums runs on the host pool only when asked to (the prover does for the H MSM,
    // the last item on a proof's critical path; for MSMs whose combine overlaps GPU work, waking the pool only costs)
    bool                              parallel_combine = false;
};
// Scalar classes of one scalar vector (msm_classes.hip; include/k16.h k16_scalar_classes_*): per mask set and bit b < 8 the
// list of wires whose scalar is below 256, has bit b set and whose table row is not (0,0); the scalars of 256 and above
// ("wide") compacted into an array of their own with their wire numbers.
struct k16_scalar_classes {
    static constexpr int MAX_SETS = 4, BITS = 8;
    k16_ctx*   ctx      = nullptr;
    uint64_t   cap_n    = 0;        // scalars it can classify
    int        max_sets = 0, n_sets = 0;
    uint64_t   n        = 0;        // scalars of the last build
    uint64_t   n_wide   = 0;        // rows of d_wide_scalars the wide MSM runs over (exact count, or the caller's bound)
    uint32_t*  d_cnt    = nullptr;  // [MAX_SETS * BITS] list lengths | [32] wide count
    uint32_t*  d_lists  = nullptr;  // [max_sets * BITS][cap_n] wire numbers
    uint32_t*  d_wide_idx     = nullptr; // [cap_n]
    uint8_t*   d_wide_scalars = nullptr; // [cap_n * 32]
    uint32_t*  h_flags  = nullptr;  // pinned, device-mapped: [0] more wide scalars than announced, [1 ..] copy of d_cnt (sync build)
    uint32_t*  h_flags_dev = nullptr;
    hipEvent_t built    = nullptr;  // recorded behind the classification
};

// the PROCESS's host thread pool (created on first use; K16_HOST_THREADS wide, default 3/4 of the usable CPUs, at most 12), or
// nullptr (K16_HOST_THREADS=1, or no thread could be started): callers then loop serially
k16_host_pool* k16_ctx_pool(k16_ctx* ctx);

// Lane streams are created on first use: ROCm multiplexes a process's streams onto 4 hardware queues by default
// (GPU_MAX_HW_QUEUES), so a stream that is never used must not take one from those that are -- the prover runs lanes
// 0-2 plus its chain stream, the MSM benchmark lanes 0-3 (measured: with an idle fifth stream a proof took 8.7 instead
// of 7.7 ms).
hipStream_t k16_lane_stream(k16_ctx* ctx, int lane);
// wait for an event on the host: hipEventSynchronize (the runtime spins: lowest latency, one busy core per waiting caller), or --
// K16_OPT_YIELDING_WAITS -- hipEventQuery + short sleeps (a waiting caller costs next to no CPU; up to ~50 us later)
hipError_t k16_event_wait(k16_ctx* ctx, hipEvent_t ev);

#define K16_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return K16_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

int k16_ws_reserve(k16_ctx* ctx, k16_devbuf& b, size_t bytes);

// Exception firewall of the C ABI (include/k16.h: "never throws").  The bodies of the extern "C" entry points use std
// containers, std::string and new; whatever they throw ends here as a status code -- a C, Rust (bindgen) or ctypes caller
// cannot unwind.  std::bad_alloc -> K16_ERR_NOMEM, anything else -> K16_ERR_HIP (the "not the caller's fault" class).
template <class F>
inline int k16_guard(k16_ctx* ctx, F&& body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try {
            if (ctx) ctx->err = "out of host memory";
        } catch (...) {
        }
        return K16_ERR_NOMEM;
    } catch (const std::exception& e) {
        try {
            if (ctx) ctx->err = std::string("internal error: ") + e.what();
        } catch (...) {
        }
        return K16_ERR_HIP;
    } catch (...) {
        return K16_ERR_HIP;
    }
}
template <class F>
inline void k16_guard_void(F&& body) noexcept
{
    try {
        body();
    } catch (...) {
    }
}

void k16_stats_begin(k16_ctx* ctx, const char* name, hipStream_t st);
void k16_stats_end(k16_ctx* ctx, hipStream_t st);
int  k16_stats_resolve(k16_ctx* ctx);

// brackets a group of launches with a HIP event pair on the context's stream (no host sync)
struct k16_stat_scope {
    k16_ctx*    ctx;
    bool        on;
    hipStream_t st;
    k16_stat_scope(k16_ctx* c, const char* n, hipStream_t s = nullptr)
        : ctx(c), on((c->stats_on == 1 || (c->stats_on == 2 && n[4] == 'a')) && !c->capturing), st(s ? s : c->stream)
    {
        if (on) k16_stats_begin(ctx, n, st);
    }
    ~k16_stat_scope()
    {
        if (on) k16_stats_end(ctx, st);
    }
};

// host-side helpers implemented in ntt.hip
int k16_ntt_get_table(k16_ctx* ctx, uint64_t max_domain, k16_ntt_table** out);
// packed9: bit 0 = data is in the packed R' domain, bit 1 = input already bit-reversed, bit 2 = skip the inverse tail
int k16_ntt_enqueue(k16_ctx* ctx, k16::Fr* d_a, uint64_t n, k16_ntt_table* tab, int inverse, hipStream_t st, int packed9);
int k16_ntt_coset_chain(k16_ctx* ctx, k16::Fr* const* src, k16::Fr* const* dst, int count, uint64_t n, k16_ntt_table* tab,
                        const k16::Fr* shift9, hipStream_t st);
int k16_ntt_build_coset_shift(k16_ctx* ctx, k16_ntt_table* tab, uint64_t n, k16::Fr** out, hipStream_t st);
