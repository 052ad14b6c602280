// tools/lab/pack_bench.cpp -- where the 0.24 ms of the witness hand-off go (host only): fork-join latency of the library's
// host pool, the scalar scan of prover.hip's WitnessPacker, an AVX2 scan, per pool width.
//   g++ -O3 -mavx2 -std=c++17 -pthread -I keyless-zk-proofs_amd/csrc tools/lab/pack_bench.cpp -o tools/lab/pack_bench
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "host_pool.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void scan_scalar(const uint8_t* src, uint8_t* narrow, uint64_t lo, uint64_t hi, uint32_t* wide)
{
    uint32_t c = 0;
    for (uint64_t i = lo; i < hi; i++) {
        uint64_t w[4];
        memcpy(w, src + i * 32, 32);
        if (((w[0] >> 8) | w[1] | w[2] | w[3]) == 0) narrow[i] = (uint8_t)w[0];
        else { narrow[i] = 0; c++; }
    }
    *wide = c;
}
static void scan_avx2(const uint8_t* src, uint8_t* narrow, uint64_t lo, uint64_t hi, uint32_t* wide)
{
    const __m256i mask = _mm256_set_epi64x(-1, -1, -1, ~0xffll);
    uint32_t c = 0;
    for (uint64_t i = lo; i < hi; i++) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(src + i * 32));
        if (_mm256_testz_si256(v, mask)) narrow[i] = src[i * 32];
        else { narrow[i] = 0; c++; }
    }
    *wide = c;
}

int main(int argc, char** argv)
{
    const uint32_t n = 1343588, T = 32, reps = 40;
    std::vector<std::vector<uint8_t>> wit(4, std::vector<uint8_t>((size_t)n * 32, 0));
    for (auto& w : wit)
        for (uint32_t i = 0; i < n; i++) {
            w[(size_t)i * 32] = (uint8_t)(i * 2654435761u >> 24);
            if (i % 50 == 7) w[(size_t)i * 32 + 9] = 1;
        }
    std::vector<uint8_t> narrow(n);
    for (int width : {4, 8, 12, 16, 24}) {
        k16_host_pool pool(width - 1);
        std::vector<uint32_t> cnt(T);
        double t_empty = 0, t_s = 0, t_v = 0;
        for (uint32_t r = 0; r < reps; r++) {
            std::this_thread::sleep_for(std::chrono::milliseconds(5)); // the pool is idle between proofs
            double t0 = now_us();
            pool.run(T, [&](unsigned) {});
            t_empty += now_us() - t0;
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            const uint8_t* src = wit[r & 3].data();
            t0 = now_us();
            pool.run(T, [&](unsigned t) { scan_scalar(src, narrow.data(), (uint64_t)n * t / T, (uint64_t)n * (t + 1) / T, &cnt[t]); });
            t_s += now_us() - t0;
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            src = wit[(r + 1) & 3].data();
            t0  = now_us();
            pool.run(T, [&](unsigned t) { scan_avx2(src, narrow.data(), (uint64_t)n * t / T, (uint64_t)n * (t + 1) / T, &cnt[t]); });
            t_v += now_us() - t0;
        }
        printf("pool width %2d: empty job of %u tasks %6.1f us   scalar scan %6.1f us   AVX2 scan %6.1f us\n", width, T, t_empty / reps, t_s / reps, t_v / reps);
    }
    // one thread alone, no pool
    uint32_t c;
    double t0 = now_us();
    scan_scalar(wit[0].data(), narrow.data(), 0, n, &c);
    double t1 = now_us();
    scan_avx2(wit[1].data(), narrow.data(), 0, n, &c);
    double t2 = now_us();
    printf("one thread, whole witness: scalar %.0f us, AVX2 %.0f us\n", t1 - t0, t2 - t1);
    return 0;
}
