// msm_sharded.hip -- ONE multi-scalar multiplication sharded over several GPUs, from C / C++ (BASELINE config 5).
//
// Replaces, for MSMs above one device's comfort (~2^24 points), what the reference does on the CPU with
// ParallelMultiexp::multiexp (RS/multiexp.cpp:183-245: the point / scalar arrays cut into nThreads chunks per window, the
// per-thread accumulators packed at the end).  Here the cut is by INDEX RANGE over devices: shard r owns the contiguous rows
// [lo_r, hi_r) of the (static) point table and of the scalar array, runs the whole device pipeline on them, and the only
// thing that ever leaves a device is the shard's partial result -- ONE XYZZ point, 128 B (G1) / 256 B (G2) -- which is
// folded with EC additions (RS/curve.cpp:91-166).  No bucket-level data crosses xGMI.
//
// Two ways to run it, both behind include/k16.h:
//   * k16_msm_sharded_*: ONE process drives all devices (one context per entry of `devices`; entries may repeat, which is
//     how the one-GPU test boxes run two shards).  A host thread per shard uploads that shard's scalars, runs k16_msm on its
//     context (chunks of 2^24 points on two lanes) and leaves the partial in host memory -- where the Horner combine of
//     every MSM ends anyway -- so the fold needs no collective at all.
//   * k16_rank_comm_*: one PROCESS per GPU (the launcher model of bench.py / torch.distributed, and of a service that runs
//     one prover process per device): the ranks exchange their partials with ONE ncclAllGather over xGMI (RCCL has no
//     elliptic-curve reduction operator, so it is a gather of raw bytes + the fold on every rank).  RCCL is loaded with
//     dlopen at the first use: libk16.so has no link-time dependency on it and a single-GPU service never maps it.
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <memory>
#include "ctx.h"

using namespace k16;

namespace {
inline void shard_range(uint64_t n, unsigned world, unsigned rank, uint64_t* lo, uint64_t* hi)
{
    // contiguous [lo, hi); the first n % world shards get one extra row (keyless-zk-proofs_amd/sharding.py: shard_range)
    const uint64_t base = n / world, extra = n % world;
    *lo = (uint64_t)rank * base + std::min<uint64_t>(rank, extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}
inline size_t aff_bytes(int group) { return group == K16_G1 ? sizeof(G1Aff) : sizeof(G2Aff); }
inline size_t xyzz_bytes(int group) { return group == K16_G1 ? sizeof(G1Xyzz) : sizeof(G2Xyzz); }
} // namespace

struct k16_msm_shards {
    int      group = K16_G1;
    uint64_t n     = 0;
    struct Shard {
        int      device = 0;
        k16_ctx* ctx    = nullptr;
        uint64_t lo = 0, hi = 0;
        void*    d_bases    = nullptr; // this shard's rows in the kernels' prepared layout (k16_msm_bases_prepare)
        void*    d_scalars  = nullptr;
        bool     have_bases = false;
        int      rc         = K16_OK;
    };
    std::vector<Shard> shards;
    std::string        err;
    double             last_ms[3] = {0, 0, 0}; // upload + device work (max over shards), fold, total of the last run
};

extern "C" int k16_msm_sharded_create(const int* devices, int n_devices, int group, uint64_t n, k16_msm_shards** out)
{
    return k16_guard(nullptr, [&]() -> int {
        if (!devices || !out || n_devices < 1 || n_devices > 64 || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
        *out = nullptr;
        std::unique_ptr<k16_msm_shards> s(new k16_msm_shards);
        s->group = group;
        s->n     = n;
        s->shards.resize((size_t)n_devices);
        int rc = K16_OK;
        for (int r = 0; r < n_devices && !rc; r++) {
            auto& sh  = s->shards[(size_t)r];
            sh.device = devices[r];
            shard_range(n, (unsigned)n_devices, (unsigned)r, &sh.lo, &sh.hi);
            rc = k16_ctx_create(sh.device, &sh.ctx);
            const uint64_t cnt = sh.hi - sh.lo;
            if (!rc) rc = k16_dev_alloc(sh.ctx, std::max<size_t>((size_t)cnt * aff_bytes(group), 16), &sh.d_bases);
            if (!rc) rc = k16_dev_alloc(sh.ctx, std::max<size_t>((size_t)cnt * 32, 16), &sh.d_scalars);
        }
        if (rc) {
            for (auto& sh : s->shards) {
                if (sh.ctx) {
                    if (sh.d_bases) (void)k16_dev_free(sh.ctx, sh.d_bases);
                    if (sh.d_scalars) (void)k16_dev_free(sh.ctx, sh.d_scalars);
                    k16_ctx_destroy(sh.ctx);
                }
            }
            return rc;
        }
        *out = s.release();
        return K16_OK;
    });
}

extern "C" void k16_msm_sharded_destroy(k16_msm_shards* s)
{
    if (!s) return;
    for (auto& sh : s->shards) {
        if (!sh.ctx) continue;
        (void)k16_sync(sh.ctx);
        if (sh.d_bases) (void)k16_dev_free(sh.ctx, sh.d_bases);
        if (sh.d_scalars) (void)k16_dev_free(sh.ctx, sh.d_scalars);
        k16_ctx_destroy(sh.ctx);
    }
    delete s;
}

extern "C" int k16_msm_sharded_count(const k16_msm_shards* s) { return s ? (int)s->shards.size() : 0; }

extern "C" int k16_msm_sharded_range(const k16_msm_shards* s, int shard, uint64_t* lo, uint64_t* hi)
{
    if (!s || shard < 0 || shard >= (int)s->shards.size()) return K16_ERR_ARG;
    if (lo) *lo = s->shards[(size_t)shard].lo;
    if (hi) *hi = s->shards[(size_t)shard].hi;
    return K16_OK;
}

extern "C" k16_ctx* k16_msm_sharded_ctx(k16_msm_shards* s, int shard)
{
    if (!s || shard < 0 || shard >= (int)s->shards.size()) return nullptr;
    return s->shards[(size_t)shard].ctx;
}

extern "C" const char* k16_msm_sharded_last_error(const k16_msm_shards* s) { return s ? s->err.c_str() : "null handle"; }

// run f(shard) on one host thread per shard (the calling thread takes shard 0) and collect the first failure
template <class F>
static int for_each_shard(k16_msm_shards* s, F f)
{
    std::vector<std::thread> th;
    auto                     work = [s, &f](size_t r) {
        try {
            s->shards[r].rc = f(s->shards[r]);
        } catch (const std::bad_alloc&) {
            s->shards[r].rc = K16_ERR_NOMEM;
        } catch (...) {
            s->shards[r].rc = K16_ERR_HIP;
        }
    };
    std::vector<size_t> inline_shards; // shards whose thread could not be started (std::system_error): run by the caller
    th.reserve(s->shards.size());
    for (size_t r = 1; r < s->shards.size(); r++) {
        try {
            th.emplace_back(work, r);
        } catch (...) {
            inline_shards.push_back(r);
        }
    }
    try { // (an exception on the calling thread must not skip the joins: a joinable std::thread's destructor terminates)
        s->shards[0].rc = f(s->shards[0]);
    } catch (const std::bad_alloc&) {
        s->shards[0].rc = K16_ERR_NOMEM;
    } catch (...) {
        s->shards[0].rc = K16_ERR_HIP;
    }
    for (size_t r : inline_shards) work(r);
    for (auto& t : th) t.join();
    for (size_t r = 0; r < s->shards.size(); r++) {
        if (s->shards[r].rc) {
            s->err = "shard " + std::to_string(r) + " (device " + std::to_string(s->shards[r].device) + "): " +
                     k16_last_error(s->shards[r].ctx);
            return s->shards[r].rc;
        }
    }
    return K16_OK;
}

// the point table, once: h_bases = n rows in the reference's format (affine, Montgomery, LE; zkey sections 5-9 layout), each
// shard uploads its slice and converts it to the kernels' row layout in place
extern "C" int k16_msm_sharded_set_bases(k16_msm_shards* s, const void* h_bases)
{
    return k16_guard(nullptr, [&]() -> int {
        if (!s || (!h_bases && s->n)) return K16_ERR_ARG;
        const size_t pb = aff_bytes(s->group);
        return for_each_shard(s, [&](k16_msm_shards::Shard& sh) -> int {
            const uint64_t cnt = sh.hi - sh.lo;
            sh.have_bases      = true;
            if (!cnt) return K16_OK;
            void* raw = nullptr; // the slice in the reference's format: converted into d_bases, then dropped
            int   rc  = k16_dev_alloc(sh.ctx, (size_t)cnt * pb, &raw);
            if (rc) return rc;
            rc = k16_h2d(sh.ctx, raw, (const char*)h_bases + sh.lo * pb, (size_t)cnt * pb);
            if (!rc) rc = k16_msm_bases_prepare(sh.ctx, s->group, raw, cnt, sh.d_bases);
            if (!rc) rc = k16_sync(sh.ctx);
            (void)k16_dev_free(sh.ctx, raw);
            return rc;
        });
    });
}

// ... or a shard's slice that is already on ITS device (reference format; copied and converted)
extern "C" int k16_msm_sharded_set_bases_device(k16_msm_shards* s, int shard, const void* d_slice)
{
    return k16_guard(nullptr, [&]() -> int {
        if (!s || shard < 0 || shard >= (int)s->shards.size()) return K16_ERR_ARG;
        auto&          sh  = s->shards[(size_t)shard];
        const uint64_t cnt = sh.hi - sh.lo;
        sh.have_bases      = true;
        if (!cnt) return K16_OK;
        if (!d_slice) return K16_ERR_ARG;
        int rc = k16_msm_bases_prepare(sh.ctx, s->group, d_slice, cnt, sh.d_bases);
        if (!rc) rc = k16_sync(sh.ctx);
        if (rc) s->err = k16_last_error(sh.ctx);
        return rc;
    });
}

static int sharded_run(k16_msm_shards* s, const void* h_scalars, const void* const* d_scalars, void* h_out_xyzz, void* h_out_affine)
{
    const size_t                   xb = xyzz_bytes(s->group);
    std::vector<unsigned char>     parts(s->shards.size() * xb);
    const auto                     t0 = std::chrono::steady_clock::now();
    for (auto& sh : s->shards)
        if (!sh.have_bases) {
            s->err = "k16_msm_sharded_run before the bases were set";
            return K16_ERR_ARG;
        }
    int rc = for_each_shard(s, [&](k16_msm_shards::Shard& sh) -> int {
        const uint64_t cnt = sh.hi - sh.lo;
        const size_t   r   = (size_t)(&sh - s->shards.data());
        const void*    ds  = d_scalars ? d_scalars[r] : sh.d_scalars;
        const size_t   pb  = aff_bytes(s->group);
        // One device call takes up to 2^24 rows (msm_api.hip); scalars that arrive in HOST memory are uploaded in pieces of
        // 2^22 rows (128 MB) so that piece i + 1 crosses PCIe while piece i is sorted and accumulated: the uploads go through
        // the context's stream (lane 0), the MSMs alternate between lanes 1 and 2.  The pieces' partial results are folded
        // like the shards' (one XYZZ point each).
        const uint64_t CH = d_scalars ? (1ull << 24) : (1ull << 22);
        if (cnt <= CH) {
            int rc2 = K16_OK;
            if (cnt && !d_scalars) rc2 = k16_h2d(sh.ctx, sh.d_scalars, (const char*)h_scalars + sh.lo * 32, (size_t)cnt * 32);
            if (!rc2) rc2 = k16_msm_enqueue_prepared(sh.ctx, s->group, sh.d_bases, ds, cnt);
            if (!rc2) rc2 = k16_msm_finish(sh.ctx, parts.data() + r * xb, nullptr);
            return rc2;
        }
        const uint64_t             chunks = (cnt + CH - 1) / CH;
        std::vector<unsigned char> cp((size_t)chunks * xb);
        uint64_t                   enq = 0, fin = 0;
        int                        rc2 = K16_OK;
        auto                       next = [&]() -> int {
            const uint64_t lo = enq * CH, c = std::min<uint64_t>(CH, cnt - lo);
            if (!d_scalars) {
                const int e = k16_h2d(sh.ctx, (char*)sh.d_scalars + lo * 32, (const char*)h_scalars + (sh.lo + lo) * 32, (size_t)c * 32);
                if (e) return e;
            }
            (void)k16_msm_set_lane(sh.ctx, d_scalars ? (int)(enq % 2) : 1 + (int)(enq % 2));
            enq++;
            return k16_msm_enqueue_prepared(sh.ctx, s->group, (const char*)sh.d_bases + lo * pb, (const char*)ds + lo * 32, c);
        };
        rc2 = next();
        while (!rc2 && fin < chunks) {
            if (enq < chunks) rc2 = next();                                  // upload + enqueue the next piece ...
            if (!rc2 && enq < chunks && enq - fin < 3) rc2 = next();         // ... and one more (two MSMs + one upload in flight)
            if (!rc2) rc2 = k16_msm_finish(sh.ctx, cp.data() + (size_t)fin * xb, nullptr);
            fin++;
        }
        (void)k16_msm_set_lane(sh.ctx, 0);
        if (rc2) {
            (void)k16_msm_abort_all(sh.ctx);
            return rc2;
        }
        return k16_points_sum(s->group, cp.data(), chunks, parts.data() + r * xb, nullptr);
    });
    const auto t1 = std::chrono::steady_clock::now();
    if (rc) return rc;
    rc            = k16_points_sum(s->group, parts.data(), s->shards.size(), h_out_xyzz, h_out_affine);
    const auto t2 = std::chrono::steady_clock::now();
    s->last_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    s->last_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    s->last_ms[2] = std::chrono::duration<double, std::milli>(t2 - t0).count();
    return rc;
}

// h_scalars: n x 32 bytes (LE integers, any 256-bit value), host memory.  Result as for k16_msm.
extern "C" int k16_msm_sharded_run(k16_msm_shards* s, const void* h_scalars, void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(nullptr, [&]() -> int {
        if (!s || (!h_scalars && s->n)) return K16_ERR_ARG;
        return sharded_run(s, h_scalars, nullptr, h_out_xyzz, h_out_affine);
    });
}
// scalars already resident: d_scalars[r] = shard r's slice on shard r's device
extern "C" int k16_msm_sharded_run_device(k16_msm_shards* s, const void* const* d_scalars, void* h_out_xyzz, void* h_out_affine)
{
    return k16_guard(nullptr, [&]() -> int {
        if (!s || !d_scalars) return K16_ERR_ARG;
        return sharded_run(s, nullptr, d_scalars, h_out_xyzz, h_out_affine);
    });
}
extern "C" int k16_msm_sharded_last_ms(const k16_msm_shards* s, double* shards_ms, double* fold_ms, double* total_ms)
{
    if (!s) return K16_ERR_ARG;
    if (shards_ms) *shards_ms = s->last_ms[0];
    if (fold_ms) *fold_ms = s->last_ms[1];
    if (total_ms) *total_ms = s->last_ms[2];
    return K16_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// One process per GPU: the partial results travel by ONE ncclAllGather (RCCL over xGMI), every rank folds.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct NcclId {
    char internal[128];
}; // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*)                                                         = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int)                                    = nullptr;
    int (*CommDestroy)(NcclComm)                                                        = nullptr;
    int (*AllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int)                                                  = nullptr;
    std::string err;
};
Rccl* rccl()
{
    static Rccl       r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.err = std::string("dlopen librccl: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        r.GetUniqueId    = (int (*)(NcclId*))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank   = (int (*)(NcclComm*, int, NcclId, int))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy    = (int (*)(NcclComm))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather      = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(r.lib, "ncclAllGather");
        r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
            r.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return &r;
}
} // namespace

struct k16_rank_comm {
    k16_ctx*    ctx  = nullptr;
    NcclComm    comm = nullptr;
    int         rank = 0, world = 1;
    hipStream_t stream = nullptr;
    void*       d_send = nullptr; // 256 B
    void*       d_recv = nullptr; // world x 256 B
    void*       h_pin  = nullptr; // (1 + world) x 256 B, page-locked
    std::string err;
};

// rank 0 calls this and hands the 128 bytes to every rank by whatever the launcher offers (a file, MPI, a TCP store)
extern "C" int k16_rank_comm_unique_id(void* out128)
{
    return k16_guard(nullptr, [&]() -> int {
        if (!out128) return K16_ERR_ARG;
        Rccl* R = rccl();
        if (!R->lib) return K16_ERR_NO_DEVICE;
        NcclId id;
        if (R->GetUniqueId(&id) != 0) return K16_ERR_HIP;
        memcpy(out128, &id, sizeof id);
        return K16_OK;
    });
}
extern "C" const char* k16_rank_comm_load_error(void) { return rccl()->err.c_str(); }

extern "C" void k16_rank_comm_destroy(k16_rank_comm* c);
extern "C" int k16_rank_comm_create(k16_ctx* ctx, int rank, int world, const void* unique_id128, k16_rank_comm** out)
{
    k16_rank_comm* c  = nullptr;
    const int      rc = k16_guard(ctx, [&]() -> int {
        if (!ctx || !out || !unique_id128 || world < 1 || rank < 0 || rank >= world) return K16_ERR_ARG;
        *out    = nullptr;
        Rccl* R = rccl();
        if (!R->lib) {
            ctx->err = R->err;
            return K16_ERR_NO_DEVICE;
        }
        K16_HIP(ctx, hipSetDevice(ctx->device));
        c        = new k16_rank_comm;
        c->ctx   = ctx;
        c->rank  = rank;
        c->world = world;
        NcclId id;
        memcpy(&id, unique_id128, sizeof id);
        const int e = R->CommInitRank(&c->comm, world, id, rank);
        if (e != 0) {
            c->comm  = nullptr;
            ctx->err = std::string("ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(e) : "error");
            return K16_ERR_HIP;
        }
        K16_HIP(ctx, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        K16_HIP(ctx, hipMalloc(&c->d_send, 256));
        K16_HIP(ctx, hipMalloc(&c->d_recv, (size_t)world * 256));
        K16_HIP(ctx, hipHostMalloc(&c->h_pin, (size_t)(1 + world) * 256, hipHostMallocDefault));
        return K16_OK;
    });
    if (rc) {
        k16_rank_comm_destroy(c); // whatever was built so far
        return rc;
    }
    *out = c;
    return K16_OK;
}

extern "C" void k16_rank_comm_destroy(k16_rank_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// every rank passes ITS shard's partial result (XYZZ, host) and receives the MSM's result: gather + EC-add fold in rank order
extern "C" int k16_rank_comm_allgather_fold(k16_rank_comm* c, int group, const void* h_partial_xyzz, void* h_out_xyzz,
                                            void* h_out_affine)
{
    k16_ctx* ctx = c ? c->ctx : nullptr;
    return k16_guard(ctx, [&]() -> int {
        if (!c || !h_partial_xyzz || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
        Rccl*        R  = rccl();
        const size_t xb = xyzz_bytes(group);
        K16_HIP(ctx, hipSetDevice(ctx->device));
        memcpy(c->h_pin, h_partial_xyzz, xb);
        K16_HIP(ctx, hipMemcpyAsync(c->d_send, c->h_pin, xb, hipMemcpyHostToDevice, c->stream));
        const int e = R->AllGather(c->d_send, c->d_recv, xb, /*ncclUint8*/ 1, c->comm, c->stream);
        if (e != 0) {
            ctx->err = std::string("ncclAllGather: ") + (R->GetErrorString ? R->GetErrorString(e) : "error");
            return K16_ERR_HIP;
        }
        char* back = (char*)c->h_pin + 256;
        K16_HIP(ctx, hipMemcpyAsync(back, c->d_recv, (size_t)c->world * xb, hipMemcpyDeviceToHost, c->stream));
        K16_HIP(ctx, hipStreamSynchronize(c->stream));
        return k16_points_sum(group, back, (uint64_t)c->world, h_out_xyzz, h_out_affine);
    });
}
