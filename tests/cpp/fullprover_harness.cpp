// tests/cpp/fullprover_harness.cpp -- what the Rust crate does through bindgen, in C++:
//   FullProver p(zkey); read p.state (rust-rapidsnark/src/lib.rs:53); r = p.prove(wtns); read r fields.
// Output (one line each): state=<int> type=<int> error=<int> ms=<int> then the JSON (if any).
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#include "k16_fullprover.hpp"

struct Peek { // mirrors the field order bindgen sees: { impl, state }
    void*           impl;
    FullProverState state;
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    int reps = argc > 3 ? atoi(argv[3]) : 1;
    FullProver p(argv[1]);
    Peek       pk;
    static_assert(sizeof(Peek) == sizeof(FullProver), "FullProver layout");
    memcpy(&pk, &p, sizeof pk);
    printf("state=%d\n", (int)pk.state);
    const int threads = argc > 4 ? atoi(argv[4]) : 1;
    if (threads <= 1) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            ProverResponse r = p.prove(argv[2]);
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
        ts.emplace_back([&]() {
            for (int i = 0; i < reps; i++) {
                ProverResponse              r = p.prove(argv[2]);
                std::lock_guard<std::mutex> lk(out_mu);
                printf("type=%d error=%d ms=%d\n", (int)r.type, (int)r.error, r.metrics.prover_time);
                printf("%s\n", r.raw_json);
            }
        });
    for (auto& t : ts) t.join();
    printf("elapsed_ms=%.3f proofs=%d\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), threads * reps);
    return 0;
}
