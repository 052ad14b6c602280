// tests/cpp/fullprover_harness.cpp -- what the Rust crate does through bindgen, in C++:
//   FullProver p(zkey); read p.state (rust-rapidsnark/src/lib.rs:53); r = p.prove(wtns); read r fields.
// Output (one line each): state=<int> type=<int> error=<int> ms=<int> then the JSON (if any).
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "k16_fullprover.hpp"
#include "k16.h"

// K16_HARNESS_MEM=1: the in-memory entry point of the same pool (k16_fullprover_prove_mem) -- the witness files are read
// once, section 2 of the iden3 container is what the prover gets
static bool read_wtns_values(const std::string& path, std::vector<unsigned char>* out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> buf;
    unsigned char              tmp[1 << 16];
    for (size_t n; (n = fread(tmp, 1, sizeof tmp, f)) > 0;) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    if (buf.size() < 12 || memcmp(buf.data(), "wtns", 4) != 0) return false;
    uint32_t nsec;
    memcpy(&nsec, &buf[8], 4);
    size_t pos = 12;
    for (uint32_t i = 0; i < nsec && pos + 12 <= buf.size(); i++) {
        uint32_t type;
        uint64_t size;
        memcpy(&type, &buf[pos], 4);
        memcpy(&size, &buf[pos + 4], 8);
        pos += 12;
        if (size > buf.size() - pos) return false;
        if (type == 2) {
            out->assign(buf.begin() + pos, buf.begin() + pos + size);
            return true;
        }
        pos += size;
    }
    return false;
}
struct MemResponse { // what the harness prints for either entry point
    int         type, error, ms;
    std::string json;
};

struct Peek { // mirrors the field order bindgen sees: { impl, state }
    void*           impl;
    FullProverState state;
};

// argv[2] may be a comma-separated list of witness files: proof number k (of a thread, or of the single-threaded loop)
// uses file (thread * reps + k) % count -- BASELINE config 4 is a wave of DISTINCT witnesses through one prover pool.
static std::vector<std::string> split_paths(const char* arg)
{
    std::vector<std::string> out;
    std::string              cur;
    for (const char* c = arg;; c++) {
        if (*c == ',' || *c == 0) {
            if (!cur.empty()) out.push_back(cur);
            cur.clear();
            if (*c == 0) break;
        } else {
            cur.push_back(*c);
        }
    }
    if (out.empty()) out.push_back("");
    return out;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    // what the service's launcher does: more hardware queues than the default 4 before the
    // process touches HIP; the library itself never changes the environment
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int reps = argc > 3 ? atoi(argv[3]) : 1;
    const std::vector<std::string> wtns = split_paths(argv[2]);
    FullProver p(argv[1]);
    Peek       pk;
    static_assert(sizeof(Peek) == sizeof(FullProver), "FullProver layout");
    memcpy(&pk, &p, sizeof pk);
    printf("state=%d\n", (int)pk.state);
    const int threads = argc > 4 ? atoi(argv[4]) : 1;
    const bool mem = getenv("K16_HARNESS_MEM") != nullptr;
    // K16_HARNESS_MEM=compact: the pool's compact hand-off (k16_fullprover_compact_lease / _prove_compact): lease a slot, write
    // the witness in the upload form into ITS pinned buffers (what a witness calculator would do while it computes the
    // wires), prove on it.  `us=` is the prove call alone (the hand-off's point is that the call no longer scans 43 MB).
    const bool compact = mem && strcmp(getenv("K16_HARNESS_MEM"), "compact") == 0;
    std::vector<std::vector<unsigned char>> values(wtns.size());
    if (mem)
        for (size_t i = 0; i < wtns.size(); i++)
            if (!read_wtns_values(wtns[i], &values[i])) values[i].clear();
    auto prove_one = [&](size_t wi) -> MemResponse {
        if (!mem) {
            ProverResponse r = p.prove(wtns[wi].c_str());
            return MemResponse{(int)r.type, (int)r.error, r.metrics.prover_time, r.raw_json};
        }
        char js[4096];
        int  ms = 0;
        int  rc = K16_ERR_FORMAT;
        if (compact && !values[wi].empty()) {
            void*     lease = nullptr;
            uint8_t * narrow = nullptr, *val = nullptr;
            uint32_t* idx = nullptr;
            uint64_t  cap = 0;
            uint32_t  nv = 0;
            rc = k16_fullprover_compact_lease(&p, &lease, &narrow, &idx, &val, &cap, &nv);
            if (rc == K16_OK) {
                const unsigned char* w = values[wi].data();
                uint64_t             n_wide = 0;
                bool                 fits = values[wi].size() / 32 >= nv;
                for (uint32_t i = 0; fits && i < nv; i++) {
                    bool wide = false;
                    for (int b = 1; b < 32; b++) wide |= w[(size_t)i * 32 + b] != 0;
                    narrow[i] = wide ? 0 : w[(size_t)i * 32];
                    if (wide) {
                        if (n_wide >= cap) {
                            fits = false;
                            break;
                        }
                        idx[n_wide] = i;
                        memcpy(val + n_wide * 32, w + (size_t)i * 32, 32);
                        n_wide++;
                    }
                }
                if (!fits) { // more wide values than the list holds: give the slot back, take the full-witness entry point
                    (void)k16_fullprover_compact_cancel(&p, lease);
                    rc = k16_fullprover_prove_mem(&p, w, values[wi].size() / 32, js, sizeof js, &ms);
                } else {
                    const auto t0 = std::chrono::steady_clock::now();
                    rc            = k16_fullprover_prove_compact(&p, lease, n_wide, js, sizeof js, &ms);
                    ms = (int)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                }
            }
        } else if (!values[wi].empty()) {
            rc = k16_fullprover_prove_mem(&p, values[wi].data(), values[wi].size() / 32, js, sizeof js, &ms);
        }
        if (rc < 0) return MemResponse{1, rc == K16_ERR_NO_DEVICE || rc == K16_ERR_HIP || rc == K16_ERR_NOMEM ? 1 : 2, ms, ""};
        return MemResponse{0, 0, ms, js};
    };
    if (threads <= 1) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            MemResponse r = prove_one(i % wtns.size());
            printf("type=%d error=%d ms=%d\n", r.type, r.error, r.ms);
            printf("%s\n", r.json.c_str());
        }
        printf("elapsed_ms=%.3f proofs=%d\n",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), reps);
        return 0;
    }
    // concurrent callers on ONE FullProver (K16_DEVICES pool behind the facade): each thread proves `reps` times
    std::mutex               out_mu;
    std::vector<std::thread> ts;
    const auto               t_begin = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++)
        ts.emplace_back([&, t]() {
            for (int i = 0; i < reps; i++) {
                const size_t                wi = ((size_t)t * reps + i) % wtns.size();
                MemResponse                 r  = prove_one(wi);
                std::lock_guard<std::mutex> lk(out_mu);
                printf("type=%d error=%d ms=%d wtns=%zu\n", r.type, r.error, r.ms, wi);
                printf("%s\n", r.json.c_str());
            }
        });
    for (auto& t : ts) t.join();
    printf("elapsed_ms=%.3f proofs=%d\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), threads * reps);
    return 0;
}
