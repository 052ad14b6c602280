// tests/cpp/fullprover_harness.cpp -- what the Rust crate does through bindgen, in C++:
//   FullProver p(zkey); read p.state (rust-rapidsnark/src/lib.rs:53); r = p.prove(wtns); read r fields.
// Output (one line each): state=<int> type=<int> error=<int> ms=<int> then the JSON (if any).
#include <cstdio>
#include <cstring>
#include <cstdlib>
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
    for (int i = 0; i < reps; i++) {
        ProverResponse r = p.prove(argv[2]);
        printf("type=%d error=%d ms=%d\n", (int)r.type, (int)r.error, r.metrics.prover_time);
        printf("%s\n", r.raw_json);
    }
    return 0;
}
