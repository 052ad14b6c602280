// tools/lab/sortlab.hip -- stand-alone timing + checking of the MSM bucket sort (K1) without the EC kernels: includes the
// product's msm_kernels.inc and calls msm_sort_plan / msm_sort_launch exactly as msm_enqueue_t does (compiles in seconds;
// msm_g1.hip takes minutes).  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I keyless-zk-proofs_amd/csrc tools/lab/sortlab.hip \
//         -L keyless-zk-proofs_amd -lk16 -Wl,-rpath,$PWD/keyless-zk-proofs_amd -o /tmp/sortlab
//   /tmp/sortlab LOG2N C FLAT(0/1) DIST(uniform|witness) REPS [check]
#include <vector>
#include <random>
#include <algorithm>
#include "msm_kernels.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// host model of for_signed_digits / for_signed_digits_flat: (bucket key, sign, payload) of every non-zero digit
static void host_digits(const uint32_t* k, unsigned c, bool flat, uint64_t i, uint64_t n, std::vector<std::pair<uint32_t, uint32_t>>& out)
{
    const unsigned W = flat ? (257 + c - 1) / c : (256 + c - 1) / c;
    const uint32_t HALF = 1u << (c - 1), FULL = 1u << c;
    uint32_t carry = 0;
    for (unsigned w = 0; w < W; w++) {
        uint32_t raw = 0;
        const unsigned bit = w * c;
        if (bit < 256) {
            const unsigned wd = bit >> 5, sh = bit & 31;
            uint64_t v = k[wd];
            if (wd + 1 < 8) v |= (uint64_t)k[wd + 1] << 32;
            raw = (uint32_t)(v >> sh) & (FULL - 1u);
        }
        const uint32_t v = raw + carry, neg = v > HALF ? 1u : 0u;
        carry = neg;
        const uint32_t m = neg ? FULL - v : v;
        if (m) {
            const uint32_t key = flat ? (m - 1u) : ((w << (c - 1)) + m - 1u);
            const uint32_t row = flat ? (uint32_t)(w * n + i) : (uint32_t)i;
            out.push_back({key, (neg << 31) | row});
        }
    }
    if (!flat && carry) out.push_back({(W << (c - 1)), (uint32_t)i});
}

int main(int argc, char** argv)
{
    const unsigned log2n = argc > 1 ? atoi(argv[1]) : 20, c = argc > 2 ? atoi(argv[2]) : 16;
    const bool     flat = argc > 3 && atoi(argv[3]);
    const char*    dist = argc > 4 ? argv[4] : "uniform";
    const int      reps = argc > 5 ? atoi(argv[5]) : 10;
    const bool     check = argc > 6;
    const uint64_t n = 1ull << log2n;
    std::mt19937_64 rng(12345);
    std::vector<uint32_t> sc(n * 8);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t* k = &sc[i * 8];
        if (!strcmp(dist, "witness")) {
            const unsigned u = rng() % 100;
            for (int j = 0; j < 8; j++) k[j] = 0;
            if (u < 90) k[0] = rng() & 1;
            else if (u < 98) k[0] = rng() & 255;
            else { for (int j = 0; j < 8; j++) k[j] = (uint32_t)rng(); k[7] &= 0x1fffffffu; }
        } else {
            for (int j = 0; j < 8; j++) k[j] = (uint32_t)rng();
            k[7] &= 0x1fffffffu;
        }
    }
    k16_ctx* ctx = nullptr;
    if (k16_ctx_create(0, &ctx) != K16_OK) { fprintf(stderr, "no device\n"); return 1; }
    void* d_sc = nullptr;
    CK(hipMalloc(&d_sc, n * 32));
    CK(hipMemcpy(d_sc, sc.data(), n * 32, hipMemcpyHostToDevice));
    k16_ctx::Lane& L = ctx->lanes[0];
    hipStream_t    st = k16_lane_stream(ctx, 0);
    SortArgs sa;
    if (msm_sort_plan(ctx, L, d_sc, n, c, flat, false, nullptr, nullptr, &sa)) { fprintf(stderr, "plan: %s\n", ctx->err.c_str()); return 1; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&]() { if (msm_sort_launch(ctx, L, st, sa, [](const char*) {})) { fprintf(stderr, "launch: %s\n", ctx->err.c_str()); exit(1); } };
    run(); run();
    CK(hipStreamSynchronize(st));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, st));
        run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
    }
    printf("sortlab n=2^%u c=%u flat=%d dist=%s W=%u nb=%u seg=%u : sort %.3f ms (best %.3f) over %d reps\n", log2n, c, (int)flat, dist,
           sa.W, sa.nb, sa.seg, sum / reps, best, reps);
    if (check) {
        std::vector<uint32_t> off(sa.nb + 1), sorted((size_t)n * sa.W), segoff(sa.nb + 1), misc(16);
        CK(hipMemcpy(off.data(), sa.offsets, (sa.nb + 1) * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(segoff.data(), sa.seg_off, (sa.nb + 1) * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(misc.data(), sa.misc, 64, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sorted.data(), sa.sorted, (size_t)off[sa.nb] * 4, hipMemcpyDeviceToHost));
        std::vector<std::pair<uint32_t, uint32_t>> want;
        want.reserve((size_t)n * sa.W);
        for (uint64_t i = 0; i < n; i++) host_digits(&sc[i * 8], c, flat, i, n, want);
        std::sort(want.begin(), want.end());
        bool ok = want.size() == off[sa.nb] && misc[0] == want.size();
        size_t pos = 0;
        uint64_t nseg = 0;
        for (uint32_t b = 0; b < sa.nb && ok; b++) {
            size_t e = pos;
            while (e < want.size() && want[e].first == b) e++;
            if (off[b] != pos || off[b + 1] != e) { ok = false; fprintf(stderr, "bucket %u: offsets %u..%u want %zu..%zu\n", b, off[b], off[b + 1], pos, e); break; }
            std::vector<uint32_t> got(sorted.begin() + pos, sorted.begin() + e), w2;
            for (size_t j = pos; j < e; j++) w2.push_back(want[j].second);
            std::sort(got.begin(), got.end()); std::sort(w2.begin(), w2.end());
            if (got != w2) { ok = false; fprintf(stderr, "bucket %u: entries differ\n", b); }
            if (segoff[b] != nseg) { ok = false; fprintf(stderr, "bucket %u: seg_off %u want %llu\n", b, segoff[b], (unsigned long long)nseg); }
            nseg += (e - pos + sa.seg - 1) / sa.seg;
            pos = e;
        }
        if (ok && (segoff[sa.nb] != nseg || misc[1] != nseg)) { ok = false; fprintf(stderr, "total segments %u / %u want %llu\n", segoff[sa.nb], misc[1], (unsigned long long)nseg); }
        // segment order: a permutation of 0..nseg-1 with non-increasing lengths
        if (ok) {
            std::vector<uint32_t> order(nseg), segb(nseg);
            CK(hipMemcpy(order.data(), sa.seg_order, nseg * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(segb.data(), sa.seg_bucket, nseg * 4, hipMemcpyDeviceToHost));
            std::vector<uint8_t> seen(nseg, 0);
            uint32_t prev = 0xffffffffu;
            for (uint64_t t = 0; t < nseg && ok; t++) {
                const uint32_t s = order[t];
                if (s >= nseg || seen[s]) { ok = false; fprintf(stderr, "order[%llu] = %u repeated / out of range\n", (unsigned long long)t, s); break; }
                seen[s] = 1;
                const uint32_t b = segb[s];
                if (b >= sa.nb || segoff[b] > s || segoff[b + 1] <= s) { ok = false; fprintf(stderr, "seg_bucket[%u] = %u wrong\n", s, b); break; }
                const uint32_t k = s - segoff[b], lo = off[b] + k * sa.seg, hi = std::min(lo + sa.seg, off[b + 1]);
                if (hi - lo > prev) { ok = false; fprintf(stderr, "segment order not longest-first at %llu\n", (unsigned long long)t); }
                prev = hi - lo;
            }
        }
        printf("check: %s (%zu entries, %llu segments)\n", ok ? "OK" : "FAILED", want.size(), (unsigned long long)nseg);
        if (!ok) return 1;
    }
    return 0;
}
