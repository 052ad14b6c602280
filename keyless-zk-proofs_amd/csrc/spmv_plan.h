// spmv_plan.h -- host-side layout of the constraint matrices A | B for k_spmv (prover.hip).  Pure C++ (no HIP): compiled
// into libk16.so and, on its own, into tests/cpp/spmv_plan_check.cpp.
//
// groth16.cpp:137-156 walks the zkey's coefficient list { m, c, s, coef } and accumulates wtns[s] * coef into row c of
// matrix m under striped spinlocks.  Here the list is regrouped once at key load so that every output row has one owner:
//   * rows of <= SPMV_LONG entries, sorted by length (longest first; the empty rows form the tail; rows of one length in the
//     order of their output positions), 64 rows to a SLICE.
//     Entry k of the row in lane l of a slice sits at  slice.off + 64 k + l : a wave reads 64 consecutive coefficients per
//     step and all its lanes loop slice.len times (the length of the slice's first = longest row; shorter rows are padded
//     with entries (wire 0, coefficient 0)).
//   * longer rows (a circom Num2Bits, a wide linear combination) keep their entries contiguous at long.off and get a whole
//     wave each.
// Row id = (m == 0 ? 0 : N) + c   (groth16.cpp:147: m == 0 -> a, else b).
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace k16 {

constexpr uint32_t SPMV_LONG = 64;
struct SpmvSlice {
    uint32_t off, len; // first entry, entries per lane
};
struct SpmvLong {
    uint32_t row, off, len;
};

struct SpmvPlan {
    std::vector<SpmvSlice> slices;   // >= 1 element (dummy when there are no short rows)
    std::vector<SpmvLong>  longs;    // >= 1 element (dummy when n_long == 0)
    std::vector<uint32_t>  row_of;   // [64 * n_slices]: row of lane l of slice s at 64 s + l, 0xffffffff = no row
    std::vector<uint32_t>  pos_of;   // [n_coefs]: entry index of the i-th coefficient of the file
    uint32_t               n_slices = 0, n_long = 0;
    uint64_t               n_entries = 0; // slices (with padding) + long rows
};

// cf: n_coefs records of 44 bytes (u32 m, u32 c, u32 s, 32-byte value), possibly unaligned (groth16.hpp:33-42).
// Returns 0, or -1 for an index out of range, -2 for more than 2^32 - 1 entries.
inline int spmv_plan_build(const uint8_t* cf, uint64_t n_coefs, uint32_t N, uint32_t n_vars, SpmvPlan* out)
{
    const size_t          n_rows = 2 * (size_t)N;
    std::vector<uint32_t> len(n_rows, 0), row(n_coefs ? n_coefs : 1);
    for (uint64_t i = 0; i < n_coefs; i++) {
        uint32_t m, c, s;
        memcpy(&m, cf + i * 44, 4);
        memcpy(&c, cf + i * 44 + 4, 4);
        memcpy(&s, cf + i * 44 + 8, 4);
        if (c >= N || s >= n_vars) return -1;
        row[i] = (m == 0 ? 0 : N) + c;
        len[row[i]]++;
    }
    // short rows by length: counting sort, longest first
    std::vector<uint32_t> by_len(SPMV_LONG + 2, 0);
    out->longs.clear();
    for (size_t r = 0; r < n_rows; r++) {
        if (len[r] > SPMV_LONG)
            out->longs.push_back({(uint32_t)r, 0, len[r]});
        else
            by_len[SPMV_LONG - len[r] + 1]++;
    }
    for (size_t l = 0; l <= SPMV_LONG; l++) by_len[l + 1] += by_len[l];
    const size_t n_short  = by_len[SPMV_LONG + 1];
    const size_t n_slices = (n_short + 63) / 64;
    out->row_of.assign(std::max<size_t>(n_slices * 64, 1), 0xffffffffu);
    std::vector<uint32_t> slot_of(n_rows, 0); // short row -> position in the sorted order
    {
        // within a length class the rows follow each other in the order of their OUTPUT positions (k_spmv stores row c at
        // the bit-reversed index of c, matrix A and B side by side): the 64 stores of a slice then fall into a few KB of
        // each array instead of 64 random 32-byte places of 64 MB (measured: the kernel's stores were 100 of its 235 us)
        std::vector<uint32_t> cur(by_len.begin(), by_len.end() - 1);
        uint32_t              logN = 0;
        while ((1ull << logN) < N) logN++;
        const bool pow2 = (1ull << logN) == N;
        for (size_t i = 0; i < n_rows; i++) {
            size_t r = i;
            if (pow2 && logN) {
                uint32_t pos = (uint32_t)(i >> 1), c = 0;
                for (uint32_t b = 0; b < logN; b++) c |= ((pos >> b) & 1u) << (logN - 1 - b);
                r = (i & 1 ? (size_t)N : 0) + c;
            }
            if (len[r] <= SPMV_LONG) {
                const uint32_t q = cur[SPMV_LONG - len[r]]++;
                out->row_of[q]   = (uint32_t)r;
                slot_of[r]       = q;
            }
        }
    }
    out->slices.assign(std::max<size_t>(n_slices, 1), SpmvSlice{0, 0});
    uint64_t total = 0;
    for (size_t s = 0; s < n_slices; s++) {
        const uint32_t first = out->row_of[s * 64]; // the longest row of the slice
        out->slices[s]       = {(uint32_t)total, len[first]};
        total += (uint64_t)len[first] * 64;
    }
    std::vector<uint32_t> long_of(out->longs.empty() ? 0 : n_rows, 0);
    for (size_t k = 0; k < out->longs.size(); k++) {
        out->longs[k].off          = (uint32_t)total;
        long_of[out->longs[k].row] = (uint32_t)k;
        total += out->longs[k].len;
    }
    if (total >= (1ull << 32)) return -2;
    out->pos_of.assign(n_coefs ? n_coefs : 1, 0);
    std::vector<uint32_t> fill(n_rows, 0);
    for (uint64_t i = 0; i < n_coefs; i++) {
        const uint32_t r = row[i], k = fill[r]++;
        if (len[r] > SPMV_LONG) {
            out->pos_of[i] = out->longs[long_of[r]].off + k;
        } else {
            const uint32_t q = slot_of[r];
            out->pos_of[i]   = out->slices[q >> 6].off + (k << 6) + (q & 63);
        }
    }
    out->n_slices  = (uint32_t)n_slices;
    out->n_long    = (uint32_t)out->longs.size();
    out->n_entries = total;
    if (out->longs.empty()) out->longs.push_back({0, 0, 0});
    return 0;
}

} // namespace k16
