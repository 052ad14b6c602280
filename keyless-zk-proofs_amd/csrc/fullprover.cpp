// fullprover.cpp -- FullProver / ProverResponse (include/k16_fullprover.hpp) on top of the k16 C ABI.
// Host-only C++; mirrors the control flow of rust-rapidsnark/rapidsnark/src/fullprover.cpp:80-260.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/k16.h"
#include <time.h>
#include "../../include/k16_fullprover.hpp"

namespace {

bool log_on()
{
    const char* e = getenv("K16_LOG");
    return e && *e && *e != '0';
}
void log_line(const char* level, const char* msg)
{
    if (!log_on()) return;
    // same fields as the reference's log() (fullprover.cpp:41-78): UTC timestamp with milliseconds, level, message,
    // native_code, target -- as VALID JSON (the reference's line carries a stray quote after "native_code":"1")
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    struct tm tmv;
    gmtime_r(&ts.tv_sec, &tmv);
    char stamp[40];
    const size_t k = strftime(stamp, sizeof stamp, "%Y-%m-%dT%H:%M:%S", &tmv);
    snprintf(stamp + k, sizeof stamp - k, ".%03ldZ", ts.tv_nsec / 1000000L);
    printf("{\"timestamp\":\"%s\",\"level\":\"%s\",\"message\":\"%s\",\"native_code\":\"1\",\"target\":\"prover_service::k16\"}\n",
           stamp, level, msg);
    fflush(stdout);
}

// Devices behind one FullProver.  K16_DEVICES="0,1,2,3" puts one resident copy of the key on each listed GPU and lets
// prove() run on whichever is free; a device may be listed more than once ("0,0,0": three provers sharing one GPU, whose
// proofs overlap on it).  Default: the single device K16_DEVICE (0).  SURVEY 8(f).3: the pool lives behind the facade, so
// the Rust side only has to stop serialising prove() calls to use it.  K16_DEVICES=all: every GPU the process can see.
std::vector<int> device_list()
{
    std::vector<int> devs;
    if (const char* e = getenv("K16_DEVICES")) {
        if (strcmp(e, "all") == 0) { // one resident key per GPU of the node (BASELINE config 4: a proof per GPU per wave)
            const int n = k16_device_count();
            for (int d = 0; d < n; d++) devs.push_back(d);
            return devs; // empty without a device: the constructor then reports it
        }
        const char* p = e;
        while (*p) {
            char* end = nullptr;
            long  v   = strtol(p, &end, 10);
            if (end == p) break;
            devs.push_back((int)v);
            p = end;
            while (*p == ',' || *p == ' ') p++;
        }
    }
    if (devs.empty()) {
        const char* e = getenv("K16_DEVICE");
        devs.push_back(e ? atoi(e) : 0);
    }
    return devs;
}

} // namespace

class FullProverImpl
{
public:
    struct Slot {
        k16_ctx*    ctx    = nullptr;
        k16_prover* prover = nullptr;
        int         device = 0;
        bool        busy   = false;
        bool        dead   = false; // a device fault hit this slot and rebuilding it failed: never handed out again
    };
    std::vector<Slot>       slots;
    std::string             zkey_path; // to rebuild a slot after a device fault
    std::mutex              mu;
    std::condition_variable cv;

    ~FullProverImpl()
    {
        for (Slot& s : slots) {
            if (s.prover) k16_prover_destroy(s.prover);
            if (s.ctx) k16_ctx_destroy(s.ctx);
        }
    }
    // blocks until a prover is free; proofs of concurrent callers run on different slots.  nullptr: every slot is dead.
    Slot* acquire()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            bool any_alive = false;
            for (Slot& s : slots) {
                if (s.dead) continue;
                any_alive = true;
                if (!s.busy) {
                    s.busy = true;
                    return &s;
                }
            }
            if (!any_alive) return nullptr;
            cv.wait(lk);
        }
    }
    // A HIP error during a proof (device fault, failed allocation, lost context) leaves the slot's context in an unknown
    // state: it is torn down and rebuilt from the key file in a fresh context before anybody else gets it.  If that
    // fails too the slot is marked dead; the other slots of the pool keep serving.
    void quarantine(Slot* s)
    {
        bool ok = false;
        try { // k16_* never throw (their own firewall); this guards the few host statements around them all the same
            if (s->prover) k16_prover_destroy(s->prover);
            if (s->ctx) k16_ctx_destroy(s->ctx);
            s->prover = nullptr;
            s->ctx    = nullptr;
            // (from the key FILE, not from a sibling's resident copy: after a device fault nothing on that device is trusted)
            ok = k16_ctx_create(s->device, &s->ctx) == K16_OK && k16_prover_create(s->ctx, zkey_path.c_str(), &s->prover) == K16_OK;
            if (ok) {
                int sharing = 0;
                for (auto& o : slots) sharing += o.device == s->device;
                if (sharing > 1) (void)k16_ctx_set_option(s->ctx, K16_OPT_SHARED_GPU, 1);
                if (slots.size() > 1) (void)k16_ctx_set_option(s->ctx, K16_OPT_YIELDING_WAITS, 1);
            }
        } catch (...) {
            ok = false;
        }
        if (!ok) {
            fprintf(stderr, "k16 FullProver: device %d could not be re-initialised after a fault; slot retired\n", s->device);
            if (s->prover) k16_prover_destroy(s->prover);
            if (s->ctx) k16_ctx_destroy(s->ctx);
            s->prover = nullptr;
            s->ctx    = nullptr;
            std::lock_guard<std::mutex> lk(mu);
            s->dead = true;
        }
    }
    void release(Slot* s) noexcept
    {
        try {
            std::lock_guard<std::mutex> lk(mu);
            s->busy = false;
            if (!s->prover) s->dead = true; // a slot without a prover must never be handed out
        } catch (...) { // a failing mutex lock: still never leave the slot marked busy
            s->busy = false;
        }
        cv.notify_all(); // also wakes waiters when the last live slot died
    }
    // Scope guard of one acquired slot: whatever leaves FullProver::prove -- a return, an exception out of the few host
    // statements between acquire and release -- gives the slot back (and retires it if it lost its prover on the way).
    // Without it an exception left the slot busy for ever and, with the default one-slot pool, every later prove()
    // waiting in acquire().
    struct Lease {
        FullProverImpl* impl;
        Slot*           slot;
        Lease(FullProverImpl* i) : impl(i), slot(i->acquire()) {}
        ~Lease()
        {
            if (slot) impl->release(slot);
        }
        Lease(const Lease&)            = delete;
        Lease& operator=(const Lease&) = delete;
    };
};

char const* const ProverResponse::empty_string = "";

ProverResponse::ProverResponse(ProverError _error)
    : type(ProverResponseType::ERROR), raw_json(ProverResponse::empty_string), error(_error), metrics(ProverResponseMetrics())
{
}

ProverResponse::ProverResponse(const char* _raw_json, ProverResponseMetrics _metrics)
    : type(ProverResponseType::SUCCESS), raw_json(_raw_json), error(ProverError::NONE), metrics(_metrics)
{
}

ProverResponse::~ProverResponse()
{
    if (raw_json != empty_string) free(const_cast<char*>(raw_json));
}

FullProver::FullProver(const char* _zkeyFileName) : impl(nullptr), state(FullProverState::ZKEY_FILE_LOAD_ERROR)
{
    if (!_zkeyFileName) return;
    FullProverImpl* p = nullptr;
    // nothing may throw across the FFI (bindgen callers cannot catch): allocation failures inside the std containers
    // used here and below end as ZKEY_FILE_LOAD_ERROR
    try {
        p            = new FullProverImpl();
        p->zkey_path = _zkeyFileName;
        const std::vector<int> devs = device_list();
        if (devs.empty()) {
            fprintf(stderr, "k16 FullProver: K16_DEVICES=all but no HIP device is visible; the prover has no CPU fallback\n");
            delete p;
            return;
        }
        for (int dev : devs) {
            FullProverImpl::Slot s;
            s.device = dev;
            // the n-th prover of a device sits n placeholder streams behind the first (include/k16.h k16_ctx_create_ex: which streams
            // of provers that share a GPU take turns on a dispatch pipe; +4.6 % proofs/s for two).  K16_POOL_STREAM_OFFSET=0 disables.
            int same_dev = 0;
            for (auto& o : p->slots) same_dev += o.device == dev;
            const char* off_env = getenv("K16_POOL_STREAM_OFFSET");
            const int   offset  = (off_env ? atoi(off_env) : 1) * same_dev;
            if (k16_ctx_create_ex(dev, ((offset % 4) + 4) % 4, &s.ctx) != K16_OK) {
                fprintf(stderr, "k16 FullProver: no usable MI355X / HIP device %d; the prover has no CPU fallback\n", dev);
                delete p;
                return;
            }
            // a device listed before already holds the key: share its read-only part (one upload, one window-table build)
            const k16_prover* sibling = nullptr;
            for (auto& o : p->slots)
                if (o.device == dev && o.prover) sibling = o.prover;
            int rc = sibling ? k16_prover_create_shared(s.ctx, sibling, &s.prover) : k16_prover_create(s.ctx, _zkeyFileName, &s.prover);
            if (rc != K16_OK) {
                // fullprover.cpp:91-100 : invalid_argument -> UNSUPPORTED_ZKEY_CURVE, system_error -> ZKEY_FILE_LOAD_ERROR
                state = (rc == K16_ERR_CURVE || rc == K16_ERR_FORMAT) ? FullProverState::UNSUPPORTED_ZKEY_CURVE
                                                                      : FullProverState::ZKEY_FILE_LOAD_ERROR;
                if (rc == K16_ERR_HIP || rc == K16_ERR_NO_DEVICE) fprintf(stderr, "k16 FullProver: %s\n", k16_last_error(s.ctx));
                k16_ctx_destroy(s.ctx);
                delete p;
                return;
            }
            p->slots.push_back(s);
        }
        // entries that share a GPU prove there at the same time: throughput tuning (include/k16.h, K16_OPT_SHARED_GPU)
        for (auto& s : p->slots) {
            int sharing = 0;
            for (auto& o : p->slots) sharing += o.device == s.device;
            if (sharing > 1) (void)k16_ctx_set_option(s.ctx, K16_OPT_SHARED_GPU, 1);
            // a pool means several callers waiting for proofs at once: their waits sleep instead of spinning (include/k16.h)
            if (p->slots.size() > 1) (void)k16_ctx_set_option(s.ctx, K16_OPT_YIELDING_WAITS, 1);
        }
    } catch (...) {
        delete p;
        state = FullProverState::ZKEY_FILE_LOAD_ERROR;
        return;
    }
    impl  = p;
    state = FullProverState::OK;
}

FullProver::~FullProver()
{
    if (impl) delete impl;
}

// The pool with the witness ALREADY IN MEMORY (include/k16.h k16_fullprover_prove_mem): what a service that keeps several
// GPUs busy binds instead of prove(path), which maps and parses a 43 MB .wtns file inside every call.  Same slot lease,
// same quarantine of a slot after a device fault; the status is the C ABI's instead of a ProverResponse.
extern "C" int k16_fullprover_prove_mem(const void* fullprover, const void* wtns_values, uint64_t n_values, char* out_json,
                                        size_t cap, int* prover_time_ms)
{
    try {
        // the object's two (private) fields, in the order the header -- the reference's -- declares them; bindgen sees the same
        struct Fields {
            FullProverImpl* impl;
            FullProverState state;
        } f;
        static_assert(sizeof(Fields) == sizeof(FullProver), "FullProver layout");
        if (prover_time_ms) *prover_time_ms = 0;
        if (!fullprover || !wtns_values || !out_json) return K16_ERR_ARG;
        memcpy(&f, fullprover, sizeof f);
        const Fields* fp = &f;
        if (fp->state != FullProverState::OK || !fp->impl) return K16_ERR_NO_DEVICE; // = PROVER_NOT_READY
        int        rc;
        const auto t0 = std::chrono::steady_clock::now();
        {
            FullProverImpl::Lease lease(fp->impl);
            FullProverImpl::Slot* slot = lease.slot;
            if (!slot) return K16_ERR_NO_DEVICE; // every device of the pool has been retired
            rc = k16_prover_prove_mem(slot->prover, wtns_values, n_values, nullptr, nullptr, out_json, cap, nullptr);
            if (rc == K16_ERR_HIP || rc == K16_ERR_NO_DEVICE) fp->impl->quarantine(slot);
        }
        if (prover_time_ms)
            *prover_time_ms = (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    } catch (const std::bad_alloc&) {
        return K16_ERR_NOMEM;
    } catch (...) {
        return K16_ERR_HIP;
    }
}

namespace {
struct FullProverFields { // the object's two (private) fields, in the order the header -- the reference's -- declares them
    FullProverImpl* impl;
    FullProverState state;
};
FullProverImpl* impl_of(const void* fullprover)
{
    static_assert(sizeof(FullProverFields) == sizeof(FullProver), "FullProver layout");
    if (!fullprover) return nullptr;
    FullProverFields f;
    memcpy(&f, fullprover, sizeof f);
    return (f.state == FullProverState::OK) ? f.impl : nullptr;
}
} // namespace

// The compact hand-off through the pool (include/k16.h): lease a slot, hand out ITS pinned upload buffers, prove, release.
extern "C" int k16_fullprover_compact_lease(const void* fullprover, void** lease, uint8_t** narrow, uint32_t** wide_idx,
                                            uint8_t** wide_val, uint64_t* wide_cap, uint32_t* n_vars)
{
    try {
        if (!lease || !narrow || !wide_idx || !wide_val || !wide_cap) return K16_ERR_ARG;
        *lease = nullptr;
        FullProverImpl* impl = impl_of(fullprover);
        if (!impl) return K16_ERR_NO_DEVICE;
        FullProverImpl::Slot* slot = impl->acquire();
        if (!slot) return K16_ERR_NO_DEVICE;
        int rc = k16_prover_compact_buffers(slot->prover, narrow, wide_idx, wide_val, wide_cap);
        if (!rc && n_vars) rc = k16_prover_info(slot->prover, n_vars, nullptr, nullptr, nullptr);
        if (rc) {
            impl->release(slot);
            return rc;
        }
        *lease = slot;
        return K16_OK;
    } catch (const std::bad_alloc&) {
        return K16_ERR_NOMEM;
    } catch (...) {
        return K16_ERR_HIP;
    }
}

static FullProverImpl::Slot* leased_slot(FullProverImpl* impl, void* lease)
{
    if (!impl || !lease) return nullptr;
    for (auto& s : impl->slots)
        if (&s == lease) return s.busy ? &s : nullptr; // (a stale or foreign pointer never reaches a prover)
    return nullptr;
}

extern "C" int k16_fullprover_prove_compact(const void* fullprover, void* lease, uint64_t n_wide, char* out_json, size_t cap,
                                            int* prover_time_ms)
{
    try {
        if (prover_time_ms) *prover_time_ms = 0;
        FullProverImpl*       impl = impl_of(fullprover);
        FullProverImpl::Slot* slot = leased_slot(impl, lease);
        if (!slot) return K16_ERR_ARG;
        struct Release { // the slot goes back whatever happens below
            FullProverImpl*       impl;
            FullProverImpl::Slot* slot;
            ~Release() { impl->release(slot); }
        } rel{impl, slot};
        if (!out_json) return K16_ERR_ARG;
        const auto t0 = std::chrono::steady_clock::now();
        const int  rc = k16_prover_prove_compact(slot->prover, n_wide, nullptr, nullptr, out_json, cap, nullptr);
        if (rc == K16_ERR_HIP || rc == K16_ERR_NO_DEVICE) impl->quarantine(slot);
        if (prover_time_ms)
            *prover_time_ms = (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    } catch (const std::bad_alloc&) {
        return K16_ERR_NOMEM;
    } catch (...) {
        return K16_ERR_HIP;
    }
}

extern "C" int k16_fullprover_compact_cancel(const void* fullprover, void* lease)
{
    try {
        FullProverImpl*       impl = impl_of(fullprover);
        FullProverImpl::Slot* slot = leased_slot(impl, lease);
        if (!slot) return K16_ERR_ARG;
        impl->release(slot);
        return K16_OK;
    } catch (...) {
        return K16_ERR_HIP;
    }
}

ProverResponse FullProver::prove(const char* input) const
{
    if (state != FullProverState::OK || !impl) return ProverResponse(ProverError::PROVER_NOT_READY);
    if (!input) return ProverResponse(ProverError::INVALID_INPUT);
    try {
        log_line("INFO", "FullProver::prove begin");
        char  json[2048];
        float dev_ms = 0;
        float prove_ms = 0;
        int   rc, dev_used = -1;
        {
            FullProverImpl::Lease lease(impl);
            FullProverImpl::Slot* slot = lease.slot;
            if (!slot) return ProverResponse(ProverError::PROVER_NOT_READY); // every device of the pool has been retired
            dev_used = slot->device;
            rc = k16_prover_prove_file_timed(slot->prover, input, nullptr, nullptr, json, sizeof json, &dev_ms, &prove_ms);
            if (rc < 0 && rc != K16_ERR_CURVE && log_on())
                fprintf(stderr, "k16 FullProver::prove failed: %s\n", k16_last_error(slot->ctx));
            // A HIP failure is the DEVICE's fault, not the caller's: PROVER_NOT_READY (the service's retry / failover class,
            // RS/fullprover.cpp:117-121), and the slot goes back into the pool only after it has been rebuilt
            if (rc == K16_ERR_HIP || rc == K16_ERR_NO_DEVICE) impl->quarantine(slot);
        }
        if (rc == K16_ERR_CURVE) {
            log_line("ERROR", "witness file uses a different curve than bn128");
            return ProverResponse(ProverError::WITNESS_GENERATION_INVALID_CURVE);
        }
        if (rc == K16_ERR_HIP || rc == K16_ERR_NO_DEVICE) {
            log_line("ERROR", "device fault during prove; prover slot re-initialised");
            return ProverResponse(ProverError::PROVER_NOT_READY);
        }
        if (rc == K16_ERR_NOMEM) { // host allocation failure inside the library: the proof was cleaned up, the slot is intact
            log_line("ERROR", "out of host memory during prove");
            return ProverResponse(ProverError::PROVER_NOT_READY);
        }
        if (rc < 0) return ProverResponse(ProverError::INVALID_INPUT);
        ProverResponseMetrics m;
        // the reference's metric brackets prover->prove() only, after the witness file has been opened and checked
        // (RS/fullprover.cpp:226-231, whole milliseconds by duration_cast = truncation)
        m.prover_time = (int)prove_ms;
        if (log_on()) {
            char line[128];
            snprintf(line, sizeof line, "Time taken for Groth16 prover: %d milliseconds (device %.3f ms) on device %d",
                     m.prover_time, dev_ms, dev_used);
            log_line("INFO", line);
        }
        char* copy = strdup(json);
        if (!copy) return ProverResponse(ProverError::PROVER_NOT_READY);
        return ProverResponse(copy, m);
    } catch (...) { // std::bad_alloc / std::system_error from the pool's mutex: never across the FFI
        return ProverResponse(ProverError::PROVER_NOT_READY);
    }
}
