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
    if (threads <= 1) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            ProverResponse r = p.prove(wtns[i % wtns.size()].c_str());
            printf("type=%d error=%d ms=%d\n", (int)r.type, (int)r.error, r.metrics.prover_time);
            printf("%s\n", r.raw_json);
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
                ProverResponse              r  = p.prove(wtns[wi].c_str());
                std::lock_guard<std::mutex> lk(out_mu);
                printf("type=%d error=%d ms=%d wtns=%zu\n", (int)r.type, (int)r.error, r.metrics.prover_time, wi);
                printf("%s\n", r.raw_json);
            }
        });
    for (auto& t : ts) t.join();
    printf("elapsed_ms=%.3f proofs=%d\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), threads * reps);
    return 0;
}
