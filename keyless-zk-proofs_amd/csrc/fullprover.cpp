// fullprover.cpp -- FullProver / ProverResponse (include/k16_fullprover.hpp) on top of the k16 C ABI.
// Host-only C++; mirrors the control flow of rust-rapidsnark/rapidsnark/src/fullprover.cpp:80-260.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/k16.h"
#include "../../include/k16_fullprover.hpp"

namespace {

std::mutex g_ctx_mu;
k16_ctx*   g_ctx       = nullptr; // one device context per process (one process per GPU)
int        g_ctx_users = 0;

bool log_on()
{
    const char* e = getenv("K16_LOG");
    return e && *e && *e != '0';
}
void log_line(const char* level, const char* msg)
{
    if (!log_on()) return;
    // same shape as the reference's log() (fullprover.cpp:67-78)
    printf("{\"level\":\"%s\",\"message\":\"%s\",\"native_code\":\"1\",\"target\":\"prover_service::k16\"}\n", level, msg);
    fflush(stdout);
}

k16_ctx* acquire_ctx()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx) {
        int         dev = 0;
        const char* e   = getenv("K16_DEVICE");
        if (e) dev = atoi(e);
        if (k16_ctx_create(dev, &g_ctx) != K16_OK) {
            g_ctx = nullptr;
            return nullptr;
        }
    }
    g_ctx_users++;
    return g_ctx;
}
void release_ctx()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (g_ctx_users > 0 && --g_ctx_users == 0) {
        k16_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}

} // namespace

class FullProverImpl
{
public:
    k16_ctx*    ctx    = nullptr;
    k16_prover* prover = nullptr;
    ~FullProverImpl()
    {
        if (prover) k16_prover_destroy(prover);
        if (ctx) release_ctx();
    }
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
    FullProverImpl* p = new (std::nothrow) FullProverImpl();
    if (!p) return;
    p->ctx = acquire_ctx();
    if (!p->ctx) {
        fprintf(stderr, "k16 FullProver: no usable MI355X / HIP device; the prover has no CPU fallback\n");
        delete p;
        return;
    }
    int rc = k16_prover_create(p->ctx, _zkeyFileName, &p->prover);
    if (rc == K16_OK) {
        impl  = p;
        state = FullProverState::OK;
        return;
    }
    // fullprover.cpp:91-100 : invalid_argument -> UNSUPPORTED_ZKEY_CURVE, system_error -> ZKEY_FILE_LOAD_ERROR
    state = (rc == K16_ERR_CURVE || rc == K16_ERR_FORMAT) ? FullProverState::UNSUPPORTED_ZKEY_CURVE
                                                          : FullProverState::ZKEY_FILE_LOAD_ERROR;
    if (rc == K16_ERR_HIP || rc == K16_ERR_NO_DEVICE) fprintf(stderr, "k16 FullProver: %s\n", k16_last_error(p->ctx));
    delete p;
}

FullProver::~FullProver()
{
    if (impl) delete impl;
}

ProverResponse FullProver::prove(const char* input) const
{
    if (state != FullProverState::OK || !impl) return ProverResponse(ProverError::PROVER_NOT_READY);
    if (!input) return ProverResponse(ProverError::INVALID_INPUT);
    log_line("INFO", "FullProver::prove begin");
    char  json[2048];
    float dev_ms = 0;
    auto  t0     = std::chrono::high_resolution_clock::now();
    int   rc     = k16_prover_prove_file(impl->prover, input, nullptr, nullptr, json, sizeof json, &dev_ms);
    auto  t1     = std::chrono::high_resolution_clock::now();
    if (rc == K16_ERR_CURVE) {
        log_line("ERROR", "witness file uses a different curve than bn128");
        return ProverResponse(ProverError::WITNESS_GENERATION_INVALID_CURVE);
    }
    if (rc < 0) {
        if (log_on()) fprintf(stderr, "k16 FullProver::prove failed: %s\n", k16_last_error(impl->ctx));
        return ProverResponse(ProverError::INVALID_INPUT);
    }
    ProverResponseMetrics m;
    m.prover_time = (int)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
    if (log_on()) {
        char line[128];
        snprintf(line, sizeof line, "Time taken for Groth16 prover: %d milliseconds (device %.3f ms)", m.prover_time, dev_ms);
        log_line("INFO", line);
    }
    return ProverResponse(strdup(json), m);
}
