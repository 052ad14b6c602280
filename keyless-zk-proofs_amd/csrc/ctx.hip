// ctx.hip -- context, device memory, timers, and the batch field / point kernels used by the
// parity tests of the device arithmetic (fq_raw_generic.cpp:12-233, curve.cpp:91-458).
#include <stdio.h>
#include <stdlib.h>
#include <sched.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include "ctx.h"
#include "bn254_fq2pair.h"
#include "bn254_fq9.h"

using namespace k16;

int k16_ws_reserve(k16_ctx* ctx, k16_devbuf& b, size_t bytes)
{
    if (b.bytes >= bytes) return K16_OK;
    if (b.p) {
        K16_HIP(ctx, hipDeviceSynchronize()); // any lane may still be using the old buffer
        K16_HIP(ctx, hipFree(b.p));
        b.p     = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    K16_HIP(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    ctx->ws_gen++;
    return K16_OK;
}

// ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  One prover uses exactly four
// streams (three MSM lanes + the polynomial chain); a pool of provers on one GPU (K16_DEVICES=0,0,0) or a prover beside RCCL
// needs more, or streams of different provers serialise behind each other: 175 -> 194 proofs/s for three provers sharing an
// MI355X (profiles/r02).  The HIP runtime reads the variable when it initialises (the process's first HIP call), so this is
// an explicit call for the HOST PROGRAM's start-up, before it creates threads or touches HIP -- the library used to set the
// variable from a load-time constructor, which changed the environment of whatever process mapped it and raced with
// getenv in other threads.  A value already present in the environment wins.  INTEGRATION.md section 4.
extern "C" int k16_runtime_hw_queues(int n)
{
    if (n < 1 || n > 64) return K16_ERR_ARG;
    char buf[16];
    snprintf(buf, sizeof buf, "%d", n);
    return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? K16_OK : K16_ERR_ARG;
}

extern "C" int k16_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n < 0 ? 0 : n;
}

k16_tuning k16_tuning::from_env()
{
    k16_tuning t;
    auto on  = [](const char* n) { return getenv(n) != nullptr; };
    auto onv = [](const char* n) { const char* e = getenv(n); return e && atoi(e) != 0; };
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    t.atomic_sort        = on("K16_ATOMIC_SORT");
    t.no_fused_convert   = on("K16_NO_FUSED_CONVERT");
    t.no_staged_sort     = on("K16_NO_STAGED_SORT");
    t.fused_bins         = on("K16_FUSED_BINS");
    t.x8                 = on("K16_X8");
    t.no_l1_prefetch     = on("K16_NO_L1_PREFETCH");
    t.ntt_tail_small     = on("K16_NTT_TAIL_SMALL");
    t.ntt_unfused        = on("K16_NTT_UNFUSED");
    t.ntt_no_stage_tables = on("K16_NTT_NO_STAGE_TABLES");
    t.no_fixed_base      = on("K16_NO_FIXED_BASE");
    t.no_stream_priority = on("K16_NO_STREAM_PRIORITY");
    t.b_sort             = onv("K16_B_SORT");
    t.b_derive           = onv("K16_B_DERIVE");
    t.no_skip_zero_rows  = on("K16_NO_SKIP_ZERO_ROWS");
    t.classes            = onv("K16_CLASSES");
    t.no_warmup          = on("K16_NO_WARMUP");
    t.spmv_full          = on("K16_SPMV_FULL");
    t.fused_hscalars     = on("K16_FUSED_HSCALARS");
    t.no_split_classes   = on("K16_NO_SPLIT_CLASSES");
    t.b2_first           = on("K16_B2_FIRST");
    t.no_acc_skip        = on("K16_NO_ACC_SKIP");
    t.h_lane             = std::max(1, std::min(k16_ctx::N_LANES - 1, num("K16_H_LANE", 1)));
    t.b1_lane            = std::max(0, std::min(k16_ctx::N_LANES - 1, num("K16_B1_LANE", 0)));
    t.h_wait_first       = onv("K16_H_WAIT_FIRST");
    t.g2_acc_split       = num("K16_G2_ACC_SPLIT", 1);
    t.witness_seg        = std::max(0, std::min(1024, num("K16_WITNESS_SEG", 0)));
    t.trace              = on("K16_TRACE");
    t.trace_enq          = on("K16_TRACE_ENQ");
    t.trace_host         = on("K16_TRACE_HOST");
    t.verify_no_coop     = on("K16_VERIFY_NO_COOP");
    t.verify_coop_trace  = on("K16_VERIFY_COOP_TRACE");
    t.seg                = num("K16_SEG", 0);
    t.wsum_mlog          = num("K16_WSUM_MLOG", -1);
    t.witness_c          = num("K16_WITNESS_C", 0);
    t.ntt_tile_log       = num("K16_NTT_TILE_LOG", 0);
    t.narrow_chain       = num("K16_NARROW_CHAIN", 0);
    t.narrow_chain_g2    = num("K16_NARROW_CHAIN_G2", 0);
    if (const char* e = getenv("K16_VERIFY_COOP_MAX")) t.verify_coop_max = strtoull(e, nullptr, 10);
    return t;
}

extern "C" int k16_ctx_create(int device, k16_ctx** out) { return k16_ctx_create_ex(device, 0, out); }

extern "C" int k16_ctx_create_ex(int device, int stream_offset, k16_ctx** out)
{
    return k16_guard(nullptr, [&]() -> int {
    if (stream_offset < 0 || stream_offset > 7) return K16_ERR_ARG;
    if (!out) return K16_ERR_ARG;
    *out  = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return K16_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return K16_ERR_NO_DEVICE;
    k16_ctx* c = new k16_ctx();
    c->device  = device;
    c->tune    = k16_tuning::from_env(); // the only place the library reads its tuning switches
    // placeholder streams first (k16_ctx_create_ex; K16_STREAM_SKEW=a[,b] adds (a * n + b) mod 8 for the n-th context of the process:
    // the round-6 experiment switch): they shift the hardware queues -- and so the dispatch pipes -- this context's
    // streams land on relative to the contexts made before it
    {
        int skew = stream_offset & 7;
        if (const char* e = getenv("K16_STREAM_SKEW")) {
            static std::atomic<int> n_ctx{0};
            const char* comma = strchr(e, ',');
            skew = (skew + n_ctx.fetch_add(1) * atoi(e) + (comma ? atoi(comma + 1) : 0)) & 7;
        }
        for (int k = 0; k < skew; k++) {
            hipStream_t ph = nullptr;
            if (hipStreamCreateWithFlags(&ph, hipStreamNonBlocking) == hipSuccess) c->placeholder_streams.push_back(ph);
        }
    }
    bool lanes_ok = true;
    for (int i = 0; i < k16_ctx::N_LANES; i++)
        lanes_ok = lanes_ok && (i > 0 || hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking) == hipSuccess) &&
                   hipEventCreateWithFlags(&c->lanes[i].sort_done, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&c->lanes[i].acc_done, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&c->lanes[i].lvl1_done, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&c->lanes[i].tail_done, hipEventDisableTiming) == hipSuccess;
    if (const char* e = getenv("K16_SERIALIZE_ACC")) c->serialize_acc = atoi(e) != 0;
    if (const char* e = getenv("K16_NTT_WG_PER_CU")) c->ntt_wg_per_cu = c->ntt_wg_per_cu_default = (unsigned)std::max(1, std::min(4, atoi(e)));
    if (const char* e = getenv("K16_WSUM_MLOG_CAP")) c->wsum_mlog_cap = (unsigned)atoi(e);
    if (const char* e = getenv("K16_GRAPHS")) c->graphs_on = atoi(e) != 0;
    if (const char* e = getenv("K16_ACC_FENCE")) c->acc_fence_mode = atoi(e);
    if (const char* e = getenv("K16_ACC_GRID")) c->acc_grid_cap = (unsigned)std::max(0, atoi(e));
    if (const char* e = getenv("K16_LEAN_SORT")) c->lean_sort = atoi(e) != 0;
    if (const char* e = getenv("K16_WC_SORT")) c->wc_sort = atoi(e) != 0;
    if (const char* e = getenv("K16_ACC_DYN")) c->acc_dyn_grid = (unsigned)std::max(0, atoi(e));
    if (const char* e = getenv("K16_ACC_LDS")) c->acc_lds_bytes = (unsigned)std::min(65536, std::max(0, atoi(e)));
    if (c->graphs_on && c->acc_fence_mode != 0) {
        fprintf(stderr, "k16: K16_GRAPHS=1 ignores K16_ACC_FENCE=%d (events of that mode are not recorded on a graph replay)\n",
                c->acc_fence_mode);
        c->acc_fence_mode = 0;
    }
    c->stream = c->lanes[0].stream;
    if (!lanes_ok ||
        hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess) {
        delete c;
        return K16_ERR_NO_DEVICE;
    }
    c->pinned_bytes = k16_ctx::PEND_SLOTS * k16_ctx::SLOT_BYTES + 65536; // + debug area
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&c->pinned_dev, c->pinned, 0) != hipSuccess) {
        delete c;
        return K16_ERR_NO_DEVICE;
    }
    for (int i = 0; i < k16_ctx::PEND_SLOTS; i++)
        if (hipEventCreateWithFlags(&c->pend_ev[i], hipEventDisableTiming) != hipSuccess) {
            delete c;
            return K16_ERR_NO_DEVICE;
        }
    *out = c;
    return K16_OK;
    });
}

extern "C" void k16_ctx_destroy(k16_ctx* c)
{
    k16_guard_void([&]() {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();
    for (auto& L : c->lanes) {
        k16_devbuf* bufs[] = {&L.ws_counts, &L.ws_offsets, &L.ws_cursor, &L.ws_sorted, &L.ws_segoff, &L.ws_segbucket,
                              &L.ws_partial, &L.ws_big, &L.ws_misc, &L.ws_lvl_a, &L.ws_lvl_b, &L.ws_lvl_c,
                              &L.ws_lvl_d, &L.ws_scan, &L.ws_conv, &L.ws_narrow};
        for (auto* b : bufs)
            if (b->p) (void)hipFree(b->p);
    }
    for (auto& kv : c->ntt_tables)
        if (kv.second.roots) (void)hipFree(kv.second.roots);
    for (auto& kv : c->ntt_tables)
        if (kv.second.roots9) (void)hipFree(kv.second.roots9);
    for (auto& kv : c->ntt_tables)
        if (kv.second.stage9) (void)hipFree(kv.second.stage9);
    for (hipStream_t ph : c->placeholder_streams) (void)hipStreamDestroy(ph);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (int i = 0; i < k16_ctx::PEND_SLOTS; i++)
        if (c->pend_ev[i]) (void)hipEventDestroy(c->pend_ev[i]);
    (void)hipEventDestroy(c->ev_a);
    (void)hipEventDestroy(c->ev_b);
    for (hipEvent_t e : c->ks_pool) (void)hipEventDestroy(e);
    for (auto& L : c->lanes) {
        for (auto& kv : L.graphs)
            if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (L.sort_done) (void)hipEventDestroy(L.sort_done);
        if (L.acc_done) (void)hipEventDestroy(L.acc_done);
        if (L.lvl1_done) (void)hipEventDestroy(L.lvl1_done);
        if (L.tail_done) (void)hipEventDestroy(L.tail_done);
        if (L.stream && (&L == &c->lanes[0] || L.stream != c->lanes[0].stream)) (void)hipStreamDestroy(L.stream);
    }
    delete c;
    });
}

// Host threads of the PROCESS: three quarters of the CPUs it may use (affinity mask), at most 12, or K16_HOST_THREADS
// (K16_UPLOAD_THREADS is the older name) -- shared by every context (ctx.h).  Measured on a 16-CPU box, Keyless-shape proof:
// 1 thread (plain witness copy) p50 7.0 ms, 4 threads 6.9-7.0, 8 threads 6.5, 12 threads 6.47.  The pool lives until the
// process ends (its workers sleep on a condition variable between jobs).
static k16_host_pool* g_host_pool       = nullptr;
static bool           g_host_pool_tried = false;
k16_host_pool* k16_ctx_pool(k16_ctx*)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_host_pool || g_host_pool_tried) return g_host_pool;
    g_host_pool_tried = true;
    unsigned    want = 0;
    const char* e    = getenv("K16_HOST_THREADS");
    if (!e) e = getenv("K16_UPLOAD_THREADS");
    if (e) {
        want = (unsigned)std::max(0, std::min(32, atoi(e)));
    } else {
        cpu_set_t set;
        unsigned  n = 1;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
        want = std::max(1u, std::min(12u, n * 3 / 4));
    }
    if (want < 2) return nullptr;
    try {
        g_host_pool = new k16_host_pool(want - 1);
    } catch (...) {
        g_host_pool = nullptr;
    }
    return g_host_pool;
}
// worker threads of the process-wide pool (0 before its first use): the budget a service has to plan for is this plus the
// threads that call into the library
extern "C" int k16_host_threads(void)
{
    return g_host_pool ? (int)g_host_pool->workers.size() : 0;
}

hipStream_t k16_lane_stream(k16_ctx* ctx, int lane)
{
    k16_ctx::Lane& L = ctx->lanes[lane];
    if (!L.stream) {
        (void)hipSetDevice(ctx->device);
        if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) {
            ctx->err = "hipStreamCreate (MSM lane)";
            L.stream = ctx->stream; // degrade to lane 0's stream rather than fail: results are the same
        }
    }
    return L.stream;
}

hipError_t k16_event_wait(k16_ctx* ctx, hipEvent_t ev)
{
    if (!ctx->yielding_waits) return hipEventSynchronize(ev);
    for (unsigned spins = 0;; spins++) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (spins < 8)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(spins < 64 ? 20 : 50));
    }
}

extern "C" int k16_ctx_set_option(k16_ctx* c, int option, int value)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    switch (option) {
    case K16_OPT_GRAPHS:
        // the K16_ACC_FENCE experiments record their events between the launches of a tail; a replayed graph does not
        // (nothing on the host runs then), so the ordering those modes promise would silently be lost
        if (value && c->acc_fence_mode != 0) {
            c->err = "K16_OPT_GRAPHS cannot be combined with K16_ACC_FENCE != 0";
            return K16_ERR_ARG;
        }
        c->graphs_on = value != 0;
        return K16_OK;
    case K16_OPT_PIPELINED_MSM:
        // several MSMs in flight on different lanes, throughput over latency
        c->serialize_acc = value != 0;
        c->wsum_mlog_cap = value ? 4 : 3;
        // (the lean sort -- every sort kernel in <= 32 VGPRs, resident beside another lane's accumulation -- is NOT switched
        // on here: measured, it moves the sort under the accumulation but the step does not get shorter, the chip being
        // power-limited during a pipelined run (profiles/r03/lean_sort_and_power.md); K16_LEAN_SORT=1 selects it)
        return K16_OK;
    case K16_OPT_SHARED_GPU:
        // several provers (contexts) prove on this GPU at once: proofs per second over the latency of one.  The NTT passes
        // then keep three workgroups per CU instead of four, which leaves a SIMD the registers for a wave of another
        // prover's bucket accumulation: +2.5-3 % proofs/s with two provers, +0-0.3 ms on a proof alone
        // (profiles/r04/ab_ntt_wg_per_cu.log)
        c->ntt_wg_per_cu = value ? std::min(3u, c->ntt_wg_per_cu_default) : c->ntt_wg_per_cu_default;
        return K16_OK;
    case K16_OPT_YIELDING_WAITS:
        c->yielding_waits = value != 0;
        return K16_OK;
    default: return K16_ERR_ARG;
    }
    });
}

extern "C" const char* k16_last_error(const k16_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" void*       k16_stream(k16_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int k16_sync(k16_ctx* c)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipDeviceSynchronize());
    return K16_OK;
    });
}
extern "C" int k16_dev_alloc(k16_ctx* c, size_t bytes, void** dptr)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !dptr) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipMalloc(dptr, bytes ? bytes : 16));
    return K16_OK;
    });
}
extern "C" int k16_dev_free(k16_ctx* c, void* dptr)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipDeviceSynchronize());
    K16_HIP(c, hipFree(dptr));
    return K16_OK;
    });
}
extern "C" int k16_h2d(k16_ctx* c, void* d, const void* h, size_t bytes)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    return K16_OK;
    });
}
extern "C" int k16_host_register(k16_ctx* c, void* h, size_t bytes)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !h || !bytes) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipHostRegister(h, bytes, hipHostRegisterPortable));
    return K16_OK;
    });
}
extern "C" int k16_host_unregister(k16_ctx* c, void* h)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !h) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipHostUnregister(h));
    return K16_OK;
    });
}
extern "C" int k16_d2h(k16_ctx* c, void* h, const void* d, size_t bytes)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    return K16_OK;
    });
}
extern "C" int k16_timer_start(k16_ctx* c)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipEventRecord(c->ev_a, c->stream));
    return K16_OK;
    });
}
extern "C" int k16_timer_stop(k16_ctx* c, float* ms)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !ms) return K16_ERR_ARG;
    K16_HIP(c, hipEventRecord(c->ev_b, c->stream));
    K16_HIP(c, hipEventSynchronize(c->ev_b));
    K16_HIP(c, hipEventElapsedTime(ms, c->ev_a, c->ev_b));
    return K16_OK;
    });
}
void k16_stats_begin(k16_ctx* c, const char* name, hipStream_t st)
{
    if (c->ks_used + 2 > c->ks_pool.size()) {
        for (int i = 0; i < 64; i++) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->ks_pool.push_back(e);
        }
    }
    c->ks_pending.emplace_back(name, c->ks_used);
    (void)hipEventRecord(c->ks_pool[c->ks_used], st);
    c->ks_used += 2;
}
void k16_stats_end(k16_ctx* c, hipStream_t st)
{
    if (c->ks_pending.empty()) return;
    (void)hipEventRecord(c->ks_pool[c->ks_pending.back().second + 1], st);
}
int k16_stats_resolve(k16_ctx* c)
{
    if (c->ks_pending.empty()) return K16_OK;
    K16_HIP(c, hipDeviceSynchronize());
    for (auto& pr : c->ks_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ks_pool[pr.second], c->ks_pool[pr.second + 1]) == hipSuccess) {
            auto& s = c->stats[pr.first];
            s.launches++;
            s.total_ms += ms;
        }
    }
    c->ks_pending.clear();
    c->ks_used = 0;
    return K16_OK;
}
extern "C" int k16_kernel_stats_enable(k16_ctx* c, int on)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    int rc = k16_stats_resolve(c);
    c->stats_on = on < 0 ? 0 : (on > 2 ? 1 : on);
    return rc;
    });
}
extern "C" int k16_kernel_stats_reset(k16_ctx* c)
{
    return k16_guard(c, [&]() -> int {
    if (!c) return K16_ERR_ARG;
    int rc = k16_stats_resolve(c);
    c->stats.clear();
    return rc;
    });
}
extern "C" int k16_kernel_stats_get(k16_ctx* c, const char* name, uint64_t* launches, double* total_ms)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !name) return K16_ERR_ARG;
    int rc = k16_stats_resolve(c);
    if (rc) return rc;
    auto it = c->stats.find(name);
    if (launches) *launches = it == c->stats.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == c->stats.end() ? 0.0 : it->second.total_ms;
    return K16_OK;
    });
}

// ---------------------------------------------------------------- batch field / point kernels
template <class PR>
__global__ void k_field_op(int op, const Fp<PR>* __restrict__ a, const Fp<PR>* __restrict__ b, Fp<PR>* __restrict__ r,
                           uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<PR> x = a[i], y = b ? b[i] : Fp<PR>::zero(), z;
    switch (op) {
    case K16_OP_ADD: z = fadd(x, y); break;
    case K16_OP_SUB: z = fsub(x, y); break;
    case K16_OP_NEG: z = fneg(x); break;
    case K16_OP_MUL: z = fmul(x, y); break;
    case K16_OP_SQR: z = fsqr(x); break;
    case K16_OP_TOMONT: z = to_mont(x); break;
    case K16_OP_FROMMONT: z = from_mont(x); break;
    default: z = Fp<PR>::zero();
    }
    r[i] = z;
}
template <class F>
__global__ void k_point_op(int op, const Xyzz<F>* __restrict__ p1, const void* __restrict__ p2, Xyzz<F>* __restrict__ r,
                           uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz<F> a = p1[i], z;
    switch (op) {
    case K16_PT_ADD: z = padd(a, ((const Xyzz<F>*)p2)[i]); break;
    case K16_PT_MADD: z = padd_mixed(a, ((const Aff<F>*)p2)[i]); break;
    case K16_PT_DBL: z = pdbl(a); break;
    default: z = Xyzz<F>::zero();
    }
    r[i] = z;
}


// ---- the same batch operations on the representations the HOT kernels use (K16_FQ9 / K16_FR9 / K16_FQ2N,
// K16_G1_ENG9 / K16_G2_ENG2N): canonical values in and out, converted exactly at the edges, the operation itself on
// the unsaturated radix-2^29 field (bn254_fq9.h) / the Eng9 and Eng2n point formulas of msm_kernels.inc.
template <class C>
__device__ __forceinline__ Fq9 add_kp(Fq9 a, unsigned k)
{
    Fq9 pp;
#pragma unroll
    for (int i = 0; i < 9; i++) pp.l[i] = C::P[i];
    for (unsigned j = 0; j < k; j++) a = fadd9(a, pp);
    return a;
}
template <class C, class PR>
__device__ __forceinline__ Fq9 f9_in(const Fp<PR>& x)
{
    Fq9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = C::K_IN[i];
    return fmul9_t<C>(fq9_unpack(x.v), k); // x*R -> x*R' , < 2p
}
template <class C, class PR>
__device__ __forceinline__ Fp<PR> f9_out(const Fq9& a) // any bound <= 12p
{
    Fq9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = C::K_OUT[i];
    Fq9    v = fmul9_t<C>(a, k);
    Fp<PR> r;
    fq9_pack(r.v, v);
    cond_sub_p<PR>(r.v);
    return r;
}
template <class C, class PR>
__global__ void k_field_op9(int op, unsigned ka, unsigned kb, const Fp<PR>* __restrict__ a, const Fp<PR>* __restrict__ b,
                            Fp<PR>* __restrict__ r, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp<PR> xa = a[i], xb = b ? b[i] : Fp<PR>::zero();
    Fq9          x = add_kp<C>(f9_in<C, PR>(xa), ka), y = add_kp<C>(f9_in<C, PR>(xb), kb);
    Fp<PR>       z;
    switch (op) {
    case K16_OP_ADD: z = f9_out<C, PR>(fadd9(x, y)); break;                       // < 4 + ka + kb
    case K16_OP_SUB: z = f9_out<C, PR>(fsub9_t<C, 8>(x, y)); break;               // needs 2 + kb <= 8
    case K16_OP_NEG: z = f9_out<C, PR>(fsub9_t<C, 8>(fq9_zero(), x)); break;
    case K16_OP_MUL: z = f9_out<C, PR>(fmul9_t<C>(x, y)); break;                  // needs (2+ka)(2+kb) <= 128
    case K16_OP_SQR: z = f9_out<C, PR>(fsqr9_t<C>(x)); break;
    case K16_OP_LAZY_ADDMUL: z = f9_out<C, PR>(fmul9_t<C>(y, fadd9_lazy(add_kp<C>(x, ka), y))); break;      // ka counted twice: x + 2 ka p
    case K16_OP_LAZY_SUBMUL: z = f9_out<C, PR>(fmul9_t<C>(y, fsub9_lazy4_t<C>(add_kp<C>(x, ka), y))); break; // (2 + 2 ka + 4) * 2 <= 128
    case K16_OP_TOMONT: { // x*R: ((x/R)*R') * R^2 / R'
        Fp<PR> r2 = Fp<PR>::r2();
        Fq9    v  = fmul9_t<C>(x, fq9_unpack(r2.v));
        fq9_pack(z.v, v);
        cond_sub_p<PR>(z.v);
    } break;
    case K16_OP_FROMMONT: { // x/R: ((x/R)*R') * 1 / R'   (what fr9_to_standard does for the H scalars)
        Fq9 one = fq9_zero();
        one.l[0] = 1;
        Fq9 v = fmul9_t<C>(x, one);
        fq9_pack(z.v, v);
        cond_sub_p<PR>(z.v);
    } break;
    default: z = Fp<PR>::zero();
    }
    r[i] = z;
}
__global__ void k_field_op_fq2n(int op, const Fq2* __restrict__ a, const Fq2* __restrict__ b, Fq2* __restrict__ r, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fq2n x = fq2n_from_canonical(a[i]), y = fq2n_from_canonical(b ? b[i] : Fq2::zero()), z;
    switch (op) {
    case K16_OP_ADD: z = fadd(x, y); break;
    case K16_OP_SUB: z = fsub(x, y); break;
    case K16_OP_NEG: z = fneg(x); break;
    case K16_OP_MUL: z = fmul(x, y); break;
    case K16_OP_SQR: z = fsqr(x); break;
    default: z = Fq2n::zero();
    }
    r[i] = fq2n_to_canonical(z);
}
// the same on LANE PAIRS (bn254_fq2pair.h): element i is worked on by lanes 2i (real part) and 2i + 1 (imaginary part)
__global__ void __launch_bounds__(256) k_field_op_fq2h(int op, const Fq2* __restrict__ a, const Fq2* __restrict__ b, Fq2* __restrict__ r,
                                                       uint64_t n)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    const unsigned h = threadIdx.x & 1u;
    if (i >= n) return; // both lanes of a pair leave together
    const Fq2  xa = a[i], ya = b ? b[i] : Fq2::zero();
    const Fq2h x{fq9_from_fq(h ? xa.b : xa.a)}, y{fq9_from_fq(h ? ya.b : ya.a)};
    Fq2h       z;
    switch (op) {
    case K16_OP_ADD: z = fadd(x, y); break;
    case K16_OP_SUB: z = fsub(x, y); break;
    case K16_OP_NEG: z = fneg(x); break;
    case K16_OP_MUL: z = fmul(x, y); break;
    case K16_OP_SQR: z = fsqr(x); break;
    default: z = Fq2h::zero();
    }
    Fq* out = reinterpret_cast<Fq*>(&r[i]) + h;
    *out    = fq9_to_fq(z.v);
}
__global__ void __launch_bounds__(64) k_point_op_pair(int op, const G2Xyzz* __restrict__ p1, const void* __restrict__ p2,
                                                      G2Xyzz* __restrict__ r, uint64_t n)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    const unsigned h = threadIdx.x & 1u;
    if (i >= n) return;
    auto half = [&](const Fq2& v) { return Fq2h{fq9_from_fq(h ? v.b : v.a)}; };
    auto in2  = [&](const G2Xyzz& p) { return Xyzz<Fq2h>{half(p.x), half(p.y), half(p.zz), half(p.zzz)}; };
    Xyzz<Fq2h> a = in2(p1[i]), z;
    switch (op) {
    case K16_PT_ADD: z = padd(a, in2(((const G2Xyzz*)p2)[i])); break;
    case K16_PT_MADD: {
        const G2Aff q = ((const G2Aff*)p2)[i];
        z             = padd_mixed(a, Aff<Fq2h>{half(q.x), half(q.y)});
    } break;
    case K16_PT_DBL: z = pdbl(a); break;
    default: z = Xyzz<Fq2h>::zero();
    }
    Fq* out = reinterpret_cast<Fq*>(&r[i]) + h; // x.a x.b y.a y.b zz.a zz.b zzz.a zzz.b
    out[0]  = fq9_to_fq(z.x.v);
    out[2]  = fq9_to_fq(z.y.v);
    out[4]  = fq9_to_fq(z.zz.v);
    out[6]  = fq9_to_fq(z.zzz.v);
}
// p1 gets X += 3*ka*p, Y += ka*p (ka <= 2: X < 8p, Y < 4p, the documented bounds of a stored Xyzz9); p2 likewise with
// kb when it is an XYZZ point (an affine row must stay < 2p)
__global__ void k_point_op_eng9(int op, unsigned ka, unsigned kb, const G1Xyzz* __restrict__ p1, const void* __restrict__ p2,
                                G1Xyzz* __restrict__ r, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto in9 = [](const G1Xyzz& p, unsigned k) {
        return Xyzz9{add_kp<Fq9C>(fq9_from_fq(p.x), 3 * k), add_kp<Fq9C>(fq9_from_fq(p.y), k), fq9_from_fq(p.zz),
                     fq9_from_fq(p.zzz)};
    };
    Xyzz9 a = in9(p1[i], ka), z;
    switch (op) {
    case K16_PT_ADD: z = padd9(a, in9(((const G1Xyzz*)p2)[i], kb)); break;
    case K16_PT_MADD: z = padd_mixed9(a, aff9_from_canonical(((const G1Aff*)p2)[i])); break;
    case K16_PT_DBL: z = pdbl9(a); break;
    case K16_PT_MADD_ACC: {
        // the bucket accumulation's own addition (acc9_madd): kb bit 0 = the entry's sign (the row is subtracted), kb bit 1 =
        // the accumulator arrives with W = -Y
        Acc9 acc = Acc9::from_xyzz(a);
        if (kb & 2u) {
            acc.w   = fsub9<4>(fq9_zero(), a.y);
            acc.neg = 1u;
        }
        acc9_madd(acc, aff9_from_canonical(((const G1Aff*)p2)[i]), kb & 1u);
        z = acc.to_xyzz();
    } break;
    default: z = Xyzz9::zero();
    }
    r[i] = G1Xyzz{fq9_to_fq(z.x), fq9_to_fq(z.y), fq9_to_fq(z.zz), fq9_to_fq(z.zzz)}; // coordinate by coordinate
}
__global__ void __launch_bounds__(64) k_point_op_eng2n(int op, const G2Xyzz* __restrict__ p1, const void* __restrict__ p2,
                                                       G2Xyzz* __restrict__ r, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto in2 = [](const G2Xyzz& p) {
        return Xyzz<Fq2n>{fq2n_from_canonical(p.x), fq2n_from_canonical(p.y), fq2n_from_canonical(p.zz),
                          fq2n_from_canonical(p.zzz)};
    };
    Xyzz<Fq2n> a = in2(p1[i]), z;
    switch (op) {
    case K16_PT_ADD: z = padd(a, in2(((const G2Xyzz*)p2)[i])); break;
    case K16_PT_MADD: {
        G2Aff q = ((const G2Aff*)p2)[i];
        z = padd_mixed(a, Aff<Fq2n>{fq2n_from_canonical(q.x), fq2n_from_canonical(q.y)});
    } break;
    case K16_PT_DBL: z = pdbl(a); break;
    default: z = Xyzz<Fq2n>::zero();
    }
    r[i] = G2Xyzz{fq2n_to_canonical(z.x), fq2n_to_canonical(z.y), fq2n_to_canonical(z.zz), fq2n_to_canonical(z.zzz)};
}

// Synthetic point table: out[i] = (start + i + 1) * G, affine Montgomery -- the deterministic bases of
// SURVEY 8(d) (the reference's own MSM test uses the same family, alt_bn128_test.cpp:183-190).
template <class F>
__global__ void __launch_bounds__(64) k_synth_points(Aff<F> gen, uint64_t start, uint64_t n, Aff<F>* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k   = start + i + 1;
    Xyzz<F>  acc = Xyzz<F>::zero();
#pragma clang loop unroll(disable)
    for (int b = 63; b >= 0; b--) {
        acc = pdbl(acc);
        if ((k >> b) & 1) acc = padd_mixed(acc, gen);
    }
    out[i] = to_affine(acc);
}

// out[i] = scalars[i] * G for arbitrary 256-bit scalars (32 B little-endian, standard form): the trapdoor side of a
// synthetic Groth16 set-up (tests/valid_key_builder.py builds keys whose proofs VERIFY from known tau, alpha, beta ...)
template <class F>
__global__ void __launch_bounds__(64) k_synth_points_scalars(Aff<F> gen, const uint8_t* __restrict__ scalars, uint64_t n,
                                                             Aff<F>* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t k[32];
    for (int b = 0; b < 32; b++) k[b] = scalars[i * 32 + b];
    out[i] = to_affine(pmul_scalar(Xyzz<F>::from_aff(gen), k));
}

static G1Aff g1_generator()
{
    G1Aff g;
    g.x = Fq::one();                 // generator (1, 2)  (alt_bn128.hpp:41)
    g.y = fdbl(Fq::one());
    return g;
}
static G2Aff g2_generator()
{
    // G2 generator (alt_bn128.hpp:43-52): decimal constants converted on the host
    static const char* const G2S[4] = {
        "10857046999023057135944570762232829481370756359578518086990519993285655852781",
        "11559732032986387107991004021392285783925812861821192530917403151452391805634",
        "8495653923123431417604973247489272438418190587263600148770280649306958101930",
        "4082367875863433681332203403145435568316851327593401208105741076214120093531"};
    Fq v[4];
    Fq ten = Fq::zero();
    ten.v[0] = 10;
    ten      = to_mont(ten);
    for (int k = 0; k < 4; k++) {
        Fq acc = Fq::zero();
        for (const char* p = G2S[k]; *p; p++) {
            Fq d  = Fq::zero();
            d.v[0] = (uint32_t)(*p - '0');
            acc   = fadd(fmul(acc, ten), to_mont(d));
        }
        v[k] = acc;
    }
    return G2Aff{Fq2{v[0], v[1]}, Fq2{v[2], v[3]}};
}

extern "C" int k16_synth_points_scalars(k16_ctx* c, int group, const void* d_scalars, uint64_t n, void* d_out_affine)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !d_scalars || !d_out_affine || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    K16_HIP(c, hipSetDevice(c->device));
    unsigned grid = (unsigned)((n + 63) / 64);
    if (group == K16_G1)
        hipLaunchKernelGGL((k_synth_points_scalars<Fq>), dim3(grid), dim3(64), 0, c->stream, g1_generator(),
                           (const uint8_t*)d_scalars, n, (G1Aff*)d_out_affine);
    else
        hipLaunchKernelGGL((k_synth_points_scalars<Fq2>), dim3(grid), dim3(64), 0, c->stream, g2_generator(),
                           (const uint8_t*)d_scalars, n, (G2Aff*)d_out_affine);
    K16_HIP(c, hipGetLastError());
    return K16_OK;
    });
}

extern "C" int k16_synth_points(k16_ctx* c, int group, uint64_t start, uint64_t n, void* d_out_affine)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !d_out_affine || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    unsigned grid = (unsigned)((n + 63) / 64);
    if (group == K16_G1) {
        G1Aff g = g1_generator();
        hipLaunchKernelGGL((k_synth_points<Fq>), dim3(grid), dim3(64), 0, c->stream, g, start, n, (G1Aff*)d_out_affine);
    } else {
        G2Aff g = g2_generator();
        hipLaunchKernelGGL((k_synth_points<Fq2>), dim3(grid), dim3(64), 0, c->stream, g, start, n, (G2Aff*)d_out_affine);
    }
    K16_HIP(c, hipGetLastError());
    return K16_OK;
    });
}

extern "C" int k16_field_op_vec(k16_ctx* c, int field, int op, const void* h_a, const void* h_b, void* h_r, uint64_t n)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !h_a || !h_r || field < K16_FQ || field > K16_FQ2H) return K16_ERR_ARG;
    const unsigned ka = (op >> 8) & 15u, kb = (op >> 12) & 15u;
    op &= 0xff;
    const bool lazy = op == K16_OP_LAZY_ADDMUL || op == K16_OP_LAZY_SUBMUL;
    if (lazy && ((field != K16_FQ9 && field != K16_FR9) || kb || ka > 14 || !h_b)) return K16_ERR_ARG;
    if (!lazy && (ka || kb) && (field < K16_FQ9 || field >= K16_FQ2N || ka > 6 || kb > 6)) return K16_ERR_ARG;
    if (!lazy && op > K16_OP_FROMMONT) return K16_ERR_ARG;
    if (field >= K16_FQ2N && op > K16_OP_SQR) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    K16_HIP(c, hipSetDevice(c->device));
    void * da = nullptr, *db = nullptr, *dr = nullptr;
    size_t bytes = (size_t)n * (field >= K16_FQ2N ? 64 : 32);
    K16_HIP(c, hipMalloc(&da, bytes));
    K16_HIP(c, hipMalloc(&dr, bytes));
    K16_HIP(c, hipMemcpyAsync(da, h_a, bytes, hipMemcpyHostToDevice, c->stream));
    if (h_b) {
        K16_HIP(c, hipMalloc(&db, bytes));
        K16_HIP(c, hipMemcpyAsync(db, h_b, bytes, hipMemcpyHostToDevice, c->stream));
    }
    unsigned grid = (unsigned)((n + 255) / 256);
    if (field == K16_FQ)
        hipLaunchKernelGGL((k_field_op<FqParams>), dim3(grid), dim3(256), 0, c->stream, op, (const Fq*)da,
                           (const Fq*)db, (Fq*)dr, n);
    else if (field == K16_FR)
        hipLaunchKernelGGL((k_field_op<FrParams>), dim3(grid), dim3(256), 0, c->stream, op, (const Fr*)da,
                           (const Fr*)db, (Fr*)dr, n);
    else if (field == K16_FQ9)
        hipLaunchKernelGGL((k_field_op9<Fq9C, FqParams>), dim3(grid), dim3(256), 0, c->stream, op, ka, kb, (const Fq*)da,
                           (const Fq*)db, (Fq*)dr, n);
    else if (field == K16_FR9)
        hipLaunchKernelGGL((k_field_op9<Fr9C, FrParams>), dim3(grid), dim3(256), 0, c->stream, op, ka, kb, (const Fr*)da,
                           (const Fr*)db, (Fr*)dr, n);
    else if (field == K16_FQ2N)
        hipLaunchKernelGGL(k_field_op_fq2n, dim3(grid), dim3(256), 0, c->stream, op, (const Fq2*)da, (const Fq2*)db,
                           (Fq2*)dr, n);
    else
        hipLaunchKernelGGL(k_field_op_fq2h, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, c->stream, op, (const Fq2*)da,
                           (const Fq2*)db, (Fq2*)dr, n);
    K16_HIP(c, hipGetLastError());
    K16_HIP(c, hipMemcpyAsync(h_r, dr, bytes, hipMemcpyDeviceToHost, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(da);
    (void)hipFree(dr);
    if (db) (void)hipFree(db);
    return K16_OK;
    });
}

extern "C" int k16_point_op_vec(k16_ctx* c, int group, int op, const void* h_p1, const void* h_p2, void* h_r,
                                uint64_t n)
{
    return k16_guard(c, [&]() -> int {
    if (!c || !h_p1 || !h_r || group < K16_G1 || group > K16_G2_PAIR) return K16_ERR_ARG;
    const unsigned ka = (op >> 8) & 15u, kb = (op >> 12) & 15u;
    op &= 0xff;
    if (op == K16_PT_MADD_ACC) {
        if (group != K16_G1_ENG9 || ka > 2 || kb > 3) return K16_ERR_ARG;
    } else if ((ka || kb) && (group != K16_G1_ENG9 || ka > 2 || kb > 2 || (kb && op != K16_PT_ADD)))
        return K16_ERR_ARG;
    if (op < K16_PT_ADD || op > K16_PT_MADD_ACC) return K16_ERR_ARG;
    if (group == K16_G2_PAIR && op == K16_PT_MADD_ACC) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    K16_HIP(c, hipSetDevice(c->device));
    const bool g1 = group == K16_G1 || group == K16_G1_ENG9;
    size_t xb  = g1 ? sizeof(G1Xyzz) : sizeof(G2Xyzz);
    size_t ab  = g1 ? sizeof(G1Aff) : sizeof(G2Aff);
    size_t p2b = (op == K16_PT_MADD || op == K16_PT_MADD_ACC) ? ab : xb;
    void * d1 = nullptr, *d2 = nullptr, *dr = nullptr;
    K16_HIP(c, hipMalloc(&d1, n * xb));
    K16_HIP(c, hipMalloc(&dr, n * xb));
    K16_HIP(c, hipMemcpyAsync(d1, h_p1, n * xb, hipMemcpyHostToDevice, c->stream));
    if (h_p2 && op != K16_PT_DBL) {
        K16_HIP(c, hipMalloc(&d2, n * p2b));
        K16_HIP(c, hipMemcpyAsync(d2, h_p2, n * p2b, hipMemcpyHostToDevice, c->stream));
    }
    unsigned grid = (unsigned)((n + 63) / 64);
    if (group == K16_G1)
        hipLaunchKernelGGL((k_point_op<Fq>), dim3(grid), dim3(64), 0, c->stream, op, (const G1Xyzz*)d1, d2,
                           (G1Xyzz*)dr, n);
    else if (group == K16_G2)
        hipLaunchKernelGGL((k_point_op<Fq2>), dim3(grid), dim3(64), 0, c->stream, op, (const G2Xyzz*)d1, d2,
                           (G2Xyzz*)dr, n);
    else if (group == K16_G1_ENG9)
        hipLaunchKernelGGL(k_point_op_eng9, dim3(grid), dim3(64), 0, c->stream, op, ka, kb, (const G1Xyzz*)d1, d2,
                           (G1Xyzz*)dr, n);
    else if (group == K16_G2_ENG2N)
        hipLaunchKernelGGL(k_point_op_eng2n, dim3(grid), dim3(64), 0, c->stream, op, (const G2Xyzz*)d1, d2,
                           (G2Xyzz*)dr, n);
    else
        hipLaunchKernelGGL(k_point_op_pair, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, c->stream, op, (const G2Xyzz*)d1, d2,
                           (G2Xyzz*)dr, n);
    K16_HIP(c, hipGetLastError());
    K16_HIP(c, hipMemcpyAsync(h_r, dr, n * xb, hipMemcpyDeviceToHost, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d1);
    (void)hipFree(dr);
    if (d2) (void)hipFree(d2);
    return K16_OK;
    });
}
