// tests/cpp/spmv_plan_check.cpp -- CPU check of the SpMV layout planner (keyless-zk-proofs_amd/csrc/spmv_plan.h): emulates
// what k_spmv does with the plan (integers mod 2^61 - 1 instead of Fr) and compares every output row with the direct walk
// over the coefficient list (RS/groth16.cpp:137-156).  Prints "ok <cases>" or the first failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "spmv_plan.h"

using namespace k16;
static const uint64_t MOD = (1ull << 61) - 1;
static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % MOD); }

static int one_case(uint32_t N, uint32_t n_vars, uint64_t n_coefs, const std::vector<uint32_t>& long_rows, unsigned seed)
{
    std::mt19937_64      rng(seed);
    std::vector<uint8_t> cf(n_coefs * 44 + 1);
    std::vector<uint64_t> val(n_coefs);
    uint64_t              at = 0;
    auto put = [&](uint64_t i, uint32_t m, uint32_t c, uint32_t s, uint64_t v) {
        memcpy(&cf[i * 44], &m, 4);
        memcpy(&cf[i * 44 + 4], &c, 4);
        memcpy(&cf[i * 44 + 8], &s, 4);
        memset(&cf[i * 44 + 12], 0, 32);
        memcpy(&cf[i * 44 + 12], &v, 8);
        val[i] = v;
    };
    for (size_t k = 0; k < long_rows.size() && at + long_rows[k] <= n_coefs; k++)
        for (uint32_t j = 0; j < long_rows[k]; j++, at++) put(at, k & 1, (uint32_t)((k * 7919 + 5) % N), rng() % n_vars, rng() % MOD);
    for (; at < n_coefs; at++) put(at, rng() & 1, rng() % N, rng() % n_vars, rng() % MOD);
    std::vector<uint64_t> w(n_vars);
    for (auto& x : w) x = rng() % MOD;
    // reference: the direct walk
    std::vector<uint64_t> want(2 * (size_t)N, 0);
    for (uint64_t i = 0; i < n_coefs; i++) {
        uint32_t m, c, s;
        memcpy(&m, &cf[i * 44], 4);
        memcpy(&c, &cf[i * 44 + 4], 4);
        memcpy(&s, &cf[i * 44 + 8], 4);
        size_t r = (m == 0 ? 0 : N) + c;
        want[r]  = (want[r] + mulmod(w[s], val[i])) % MOD;
    }
    SpmvPlan plan;
    if (spmv_plan_build(cf.data(), n_coefs, N, n_vars, &plan)) return 1;
    // entries as prover.hip fills them
    std::vector<uint32_t> wire(plan.n_entries ? plan.n_entries : 1, 0);
    std::vector<uint64_t> coef(plan.n_entries ? plan.n_entries : 1, 0);
    std::vector<uint8_t>  used(plan.n_entries ? plan.n_entries : 1, 0);
    for (uint64_t i = 0; i < n_coefs; i++) {
        uint32_t pos = plan.pos_of[i], s;
        if (pos >= plan.n_entries || used[pos]) return 2; // every coefficient has its own entry
        used[pos] = 1;
        memcpy(&s, &cf[i * 44 + 8], 4);
        wire[pos] = s;
        coef[pos] = val[i];
    }
    // what k_spmv computes
    std::vector<uint64_t> got(2 * (size_t)N, ~0ull);
    size_t                rows_seen = 0;
    uint32_t              prev_len  = ~0u;
    for (uint32_t s = 0; s < plan.n_slices; s++) {
        const SpmvSlice sl = plan.slices[s];
        if (sl.len > SPMV_LONG || sl.len > prev_len) return 3; // sorted, longest first
        prev_len = sl.len;
        for (uint32_t lane = 0; lane < 64; lane++) {
            uint64_t acc = 0;
            for (uint32_t k = 0; k < sl.len; k++) {
                const uint64_t e = (uint64_t)sl.off + 64ull * k + lane;
                if (e >= plan.n_entries) return 4;
                acc = (acc + mulmod(w[wire[e]], coef[e])) % MOD;
            }
            const uint32_t row = plan.row_of[64ull * s + lane];
            if (row == 0xffffffffu) continue;
            if (row >= 2 * (size_t)N || got[row] != ~0ull) return 5; // every row exactly once
            got[row] = acc;
            rows_seen++;
        }
    }
    for (uint32_t k = 0; k < plan.n_long; k++) {
        const SpmvLong L = plan.longs[k];
        if (L.len <= SPMV_LONG || got[L.row] != ~0ull) return 6;
        uint64_t acc = 0;
        for (uint32_t j = 0; j < L.len; j++) acc = (acc + mulmod(w[wire[L.off + j]], coef[L.off + j])) % MOD;
        got[L.row] = acc;
        rows_seen++;
    }
    if (rows_seen != 2 * (size_t)N) return 7;
    // rows of one length follow each other in the order of their output positions (k_spmv stores row c of matrix m at the
    // bit-reversed index of c; position-major, A before B): the stores of a slice stay within a few KB
    {
        uint32_t logN = 0;
        while ((1ull << logN) < N) logN++;
        std::vector<uint32_t> len(2 * (size_t)N, 0);
        for (uint64_t i = 0; i < n_coefs; i++) {
            uint32_t m, c;
            memcpy(&m, &cf[i * 44], 4);
            memcpy(&c, &cf[i * 44 + 4], 4);
            len[(m == 0 ? 0 : N) + c]++;
        }
        auto key = [&](uint32_t row) -> uint64_t {
            const uint32_t c = row < N ? row : row - N;
            uint32_t       pos = 0;
            for (uint32_t b = 0; b < logN; b++) pos |= ((c >> b) & 1u) << (logN - 1 - b);
            return 2ull * pos + (row < N ? 0 : 1);
        };
        uint32_t prev_row = 0xffffffffu;
        for (size_t q = 0; q < 64ull * plan.n_slices; q++) {
            const uint32_t row = plan.row_of[q];
            if (row == 0xffffffffu) continue;
            if (prev_row != 0xffffffffu && len[prev_row] == len[row] && key(prev_row) >= key(row)) return 10;
            prev_row = row;
        }
    }
    for (size_t r = 0; r < 2 * (size_t)N; r++)
        if (got[r] != want[r]) return 8;
    // padding stays small: at most one slice-worth per distinct length
    if (plan.n_entries > n_coefs + 64ull * (SPMV_LONG + 1) * 64) return 9;
    return 0;
}

int main()
{
    struct Case {
        uint32_t N, n_vars;
        uint64_t n_coefs;
        std::vector<uint32_t> longs;
    } cases[] = {
        {1, 3, 0, {}}, {1, 3, 5, {}}, {2, 4, 1, {}}, {8, 10, 40, {}}, {64, 100, 64, {}}, {64, 100, 5000, {65, 64, 63}},
        {1024, 3000, 2500, {200, 1000}}, {4096, 9000, 30000, {65, 200, 1000, 5000, 64, 63, 129}}, {1 << 14, 40000, 100000, {70000}},
        {32, 5, 4000, {}},  // every row long
    };
    int n = 0;
    for (const Case& c : cases) {
        for (unsigned seed = 1; seed <= 3; seed++, n++) {
            int rc = one_case(c.N, c.n_vars, c.n_coefs, c.longs, seed * 977 + n);
            if (rc) {
                printf("FAIL case %d (N=%u n_coefs=%llu) rc=%d\n", n, c.N, (unsigned long long)c.n_coefs, rc);
                return 1;
            }
        }
    }
    // an index out of range is refused
    {
        uint8_t  rec[44] = {0};
        uint32_t c = 8;
        memcpy(rec + 4, &c, 4);
        SpmvPlan plan;
        if (spmv_plan_build(rec, 1, 8, 4, &plan) != -1) {
            printf("FAIL: constraint index out of range accepted\n");
            return 1;
        }
    }
    printf("ok %d\n", n);
    return 0;
}
