// tools/lab/affinelab.hip -- MEASUREMENT of the arithmetic alternative to the product's XYZZ bucket accumulation: batched-affine
// pair additions (one shared inversion per workgroup through Montgomery's trick), on the product's own radix-2^29 field.
// Not part of the library; answers "what would an accumulation built on affine + affine -> affine cost per addition on this
// chip" with a number instead of a cost model.
//
//   out[i] = A[ia[i]] + B[ib[i]]   for N independent pairs of affine G1 points (64-byte packed R' form, like the MSM's bases)
//
// A workgroup of 256 lanes owns K*256 pairs, lane t the pairs (blk*K + j)*256 + t:
//   phase 1  d_j = x2 - x1, running product run_j = d_0 .. d_j  -> scratch (9 coalesced dwords per pair), K-1 multiplications
//   phase 2  the 256 lane products -> their 256 inverses: inclusive prefix and suffix scans through LDS (8 + 8
//            multiplications per lane), ONE binary-GCD inversion of the total by one lane of one wave (rotating with the
//            workgroup number so that the four SIMDs share that work), 2 multiplications
//   phase 3  back sweep: inv_j = c * run_(j-1), c *= d_j ; lambda = (y2 - y1) inv_j ; x3 = lambda^2 - x1 - x2 ;
//            y3 = lambda (x1 - x3) - y1                                     (5 multiplications + 1 squaring per pair)
// i.e. 5M + 1S + (K-1)/K + 18/K multiplications per addition against 8M + 2S of the mixed XYZZ addition (bn254_fq9.h
// padd_mixed9), for ~330 bytes of memory traffic per addition against 64.
//
//   tools/lab/build_affinelab.sh && tools/lab/affinelab LOG2N K(16|32|64) GATHER(0/1) REPS [noinv]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "bn254_curve.h"
#include "bn254_fq9.h"

using namespace k16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

__device__ __forceinline__ Fq9 ld_packed(const uint4* p)
{
    uint4    a = p[0], b = p[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fq9_unpack(w);
}
__device__ __forceinline__ void st_packed(uint4* p, const Fq9& v)
{
    uint32_t w[8];
    fq9_pack(w, v);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ void st_soa(uint32_t* s, size_t n, size_t idx, const Fq9& v)
{
#pragma unroll
    for (int l = 0; l < 9; l++) s[(size_t)l * n + idx] = v.l[l];
}
__device__ __forceinline__ Fq9 ld_soa(const uint32_t* s, size_t n, size_t idx)
{
    Fq9 v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = s[(size_t)l * n + idx];
    return v;
}
// LDS exchange of whole field elements, one dword plane per limb (conflict-free)
__device__ __forceinline__ void lds_put(uint32_t* buf, unsigned t, const Fq9& v)
{
#pragma unroll
    for (int l = 0; l < 9; l++) buf[l * 256 + t] = v.l[l];
}
__device__ __forceinline__ Fq9 lds_get(const uint32_t* buf, unsigned t)
{
    Fq9 v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = buf[l * 256 + t];
    return v;
}
// x^-1 in the R' domain by the binary extended Euclid of bn254_field.h (canonical Montgomery in between)
__device__ __attribute__((noinline)) Fq9 finv9_bgcd(const Fq9& a) { return fq9_from_fq(finv_bgcd<FqParams>(fq9_to_fq(a))); }

template <int K, bool GATHER, bool NOINV>
__global__ void __launch_bounds__(256) k_pair_add(const uint4* __restrict__ pa, const uint4* __restrict__ pb,
                                                  const uint32_t* __restrict__ ia, const uint32_t* __restrict__ ib,
                                                  uint32_t* __restrict__ scratch, uint4* __restrict__ out,
                                                  uint32_t* __restrict__ flags, size_t n)
{
    __shared__ uint32_t pre[9 * 256], suf[9 * 256], inv_total[9];
    const unsigned t    = threadIdx.x;
    const size_t   base = (size_t)blockIdx.x * K * 256 + t;
    // ---- phase 1
    Fq9 run;
#pragma clang loop unroll(disable)
    for (int j = 0; j < K; j++) {
        const size_t   idx = base + (size_t)j * 256;
        const uint32_t a = GATHER ? ia[idx] : (uint32_t)idx, b = GATHER ? ib[idx] : (uint32_t)idx;
        Fq9            d = fsub9<2>(ld_packed(pb + (size_t)b * 4), ld_packed(pa + (size_t)a * 4)); // < 4p
        if (fq9_is_zero_mod_p<4>(d)) { // x1 == x2: doubling or P + (-P) -- a slow path in a real accumulation; counted here
            d = fq9_one();
            atomicAdd(flags, 1u);
        }
        run = j == 0 ? d : fmul9(run, d); // 2 * 4 -> < 2p
        st_soa(scratch, n, idx, run);
    }
    // ---- phase 2: inverse of every lane's product
    Fq9 p = run, s = run;
    lds_put(pre, t, p);
    lds_put(suf, t, s);
    __syncthreads();
#pragma clang loop unroll(disable)
    for (unsigned o = 1; o < 256; o <<= 1) {
        Fq9  pl, sr;
        bool hp = t >= o, hs = t + o < 256;
        if (hp) pl = lds_get(pre, t - o);
        if (hs) sr = lds_get(suf, t + o);
        __syncthreads();
        if (hp) p = fmul9(p, pl);
        if (hs) s = fmul9(s, sr);
        lds_put(pre, t, p);
        lds_put(suf, t, s);
        __syncthreads();
    }
    if (t == 64 * (blockIdx.x & 3)) { // one lane, in a wave that rotates with the workgroup number
        Fq9 tot = lds_get(pre, 255);
        Fq9 it  = NOINV ? tot : finv9_bgcd(tot);
#pragma unroll
        for (int l = 0; l < 9; l++) inv_total[l] = it.l[l];
    }
    __syncthreads();
    Fq9 c;
#pragma unroll
    for (int l = 0; l < 9; l++) c.l[l] = inv_total[l];
    if (t > 0) c = fmul9(c, lds_get(pre, t - 1));
    if (t < 255) c = fmul9(c, lds_get(suf, t + 1));
    // ---- phase 3: back sweep
#pragma clang loop unroll(disable)
    for (int j = K - 1; j >= 0; j--) {
        const size_t   idx = base + (size_t)j * 256;
        const uint32_t a = GATHER ? ia[idx] : (uint32_t)idx, b = GATHER ? ib[idx] : (uint32_t)idx;
        const uint4 *  qa = pa + (size_t)a * 4, *qb = pb + (size_t)b * 4;
        Fq9            x1 = ld_packed(qa), y1 = ld_packed(qa + 2), x2 = ld_packed(qb), y2 = ld_packed(qb + 2);
        Fq9            d  = fsub9<2>(x2, x1);
        const bool     exc = fq9_is_zero_mod_p<4>(d);
        if (exc) d = fq9_one();
        Fq9 inv_j = j > 0 ? fmul9(c, ld_soa(scratch, n, idx - 256)) : c; // < 2p
        c         = fmul9(c, d);
        Fq9 lam   = fmul9(fsub9<2>(y2, y1), inv_j);                       // 4 * 2 -> < 2p
        Fq9 x3    = fsub9<2>(fsub9<2>(fsqr9(lam), x1), x2);               // < 2 + 2 + 2 = 6p
        Fq9 y3    = fsub9<2>(fmul9(lam, fsub9<8>(x1, x3)), y1);           // 2 * 10 -> < 2p ; - y1 -> < 4p
        if (exc) x3 = y3 = fq9_zero();
        st_packed(out + idx * 4, fred9(x3));
        st_packed(out + idx * 4 + 2, fred9(y3));
    }
}

// the product's formula on the same pairs, for the same traffic pattern: affine + affine -> XYZZ (6 multiplications, no
// inversion) -- the first addition of a bucket segment; and a chain of K mixed additions per lane (what k_accumulate does)
template <int K, bool GATHER>
__global__ void __launch_bounds__(128) k_xyzz_chain(const uint4* __restrict__ pa, const uint32_t* __restrict__ ia,
                                                    uint4* __restrict__ out, size_t n)
{
    const size_t base = (size_t)blockIdx.x * K * 128 + threadIdx.x;
    Xyzz9        acc  = Xyzz9::zero();
#pragma clang loop unroll(disable)
    for (int j = 0; j < K; j++) {
        const size_t   idx = base + (size_t)j * 128;
        const uint32_t a   = GATHER ? ia[idx] : (uint32_t)idx;
        const uint4*   qa  = pa + (size_t)a * 4;
        Aff9           pt{ld_packed(qa), ld_packed(qa + 2)};
        acc = padd_mixed9(acc, pt);
    }
    const size_t o = ((size_t)blockIdx.x * 128 + threadIdx.x) * 8;
    st_packed(out + o, fred9(acc.x));
    st_packed(out + o + 2, fred9(acc.y));
    st_packed(out + o + 4, acc.zz);
    st_packed(out + o + 6, acc.zzz);
}

} // namespace

template <int K, bool G>
static float run_pair(bool noinv, const uint4* pa, const uint4* pb, const uint32_t* ia, const uint32_t* ib, uint32_t* scratch,
                      uint4* out, uint32_t* flags, size_t n, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(n / (256 * K));
    float          best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0));
        if (noinv)
            hipLaunchKernelGGL((k_pair_add<K, G, true>), dim3(grid), dim3(256), 0, 0, pa, pb, ia, ib, scratch, out, flags, n);
        else
            hipLaunchKernelGGL((k_pair_add<K, G, false>), dim3(grid), dim3(256), 0, 0, pa, pb, ia, ib, scratch, out, flags, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, ms);
    }
    return best;
}
template <int K, bool G>
static float run_chain(const uint4* pa, const uint32_t* ia, uint4* out, size_t n, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(n / (128 * K));
    float          best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_xyzz_chain<K, G>), dim3(grid), dim3(128), 0, 0, pa, ia, out, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char** argv)
{
    const unsigned log2n = argc > 1 ? atoi(argv[1]) : 22;
    const int      K     = argc > 2 ? atoi(argv[2]) : 32;
    const bool     gather = argc > 3 ? atoi(argv[3]) != 0 : true;
    const int      reps  = argc > 4 ? atoi(argv[4]) : 5;
    const bool     noinv = argc > 5 && !strcmp(argv[5], "noinv");
    const size_t   n     = (size_t)1 << log2n;
    // table of M distinct points i*G (i = 1 .. M), replicated over a 2^20-entry (64 MB) array so that gathers miss L2
    const unsigned M = 4096, TAB = 1u << 20;
    std::vector<Aff<Fq>> base(M);
    {
        Aff<Fq> g{Fq::one(), fdbl(Fq::one())};
        Xyzz<Fq> acc = Xyzz<Fq>::from_aff(g);
        for (unsigned i = 0; i < M; i++) {
            base[i] = to_affine(acc);
            acc     = padd_mixed(acc, g);
        }
    }
    std::vector<uint32_t> tab((size_t)TAB * 16);
    for (unsigned i = 0; i < TAB; i++) {
        Aff9 a = aff9_from_canonical(base[i % M]);
        // pack needs < 2^256: fq9_from_fq returns < 2p
        fq9_pack(&tab[(size_t)i * 16], a.x);
        fq9_pack(&tab[(size_t)i * 16 + 8], a.y);
    }
    std::mt19937_64       rng(7);
    std::vector<uint32_t> ia(n), ib(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t a, b;
        do {
            a = gather ? (uint32_t)(rng() % TAB) : (uint32_t)(i % TAB);
            b = (uint32_t)(rng() % TAB);
        } while (a % M == b % M);
        ia[i] = a;
        ib[i] = b;
    }
    uint4 *   d_tab, *d_out;
    uint32_t *d_ia, *d_ib, *d_scr, *d_flags;
    CK(hipMalloc(&d_tab, (size_t)TAB * 64));
    CK(hipMalloc(&d_out, n * 64));
    CK(hipMalloc(&d_ia, n * 4));
    CK(hipMalloc(&d_ib, n * 4));
    CK(hipMalloc(&d_scr, n * 36));
    CK(hipMalloc(&d_flags, 4));
    CK(hipMemset(d_flags, 0, 4));
    CK(hipMemcpy(d_tab, tab.data(), (size_t)TAB * 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ia, ia.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ib, ib.data(), n * 4, hipMemcpyHostToDevice));
    float ms = 0, msx = 0;
    // (the index arrays always drive the loads here: "gather 0" only makes the A side sequential)
    switch (K) {
    case 16: ms = run_pair<16, true>(noinv, d_tab, d_tab, d_ia, d_ib, d_scr, d_out, d_flags, n, reps); break;
    case 32: ms = run_pair<32, true>(noinv, d_tab, d_tab, d_ia, d_ib, d_scr, d_out, d_flags, n, reps); break;
    case 64: ms = run_pair<64, true>(noinv, d_tab, d_tab, d_ia, d_ib, d_scr, d_out, d_flags, n, reps); break;
    default: fprintf(stderr, "K must be 16, 32 or 64\n"); return 2;
    }
    uint32_t flags = 0;
    CK(hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost));
    printf("affinelab n=2^%u K=%d gather=%d %s: pair-add kernel %.3f ms = %.2f G additions/s (exceptional pairs seen: %u)\n", log2n, K,
           (int)gather, noinv ? "WITHOUT the inversion (upper bound of a grid-wide shared inversion)" : "one bgcd inversion per workgroup",
           ms, n / ms / 1e6, flags / (unsigned)(reps + 1));
    // check a sample against the canonical host formulas
    if (!noinv) {
        std::vector<uint32_t> out(n * 16);
        CK(hipMemcpy(out.data(), d_out, n * 64, hipMemcpyDeviceToHost));
        unsigned bad = 0, checked = 0;
        for (size_t i = 0; i < n; i += n / 4096 + 1) {
            Aff<Fq> want = to_affine(padd_mixed(Xyzz<Fq>::from_aff(base[ia[i] % M]), base[ib[i] % M]));
            Fq      gx = fq9_to_fq(fq9_unpack(&out[i * 16])), gy = fq9_to_fq(fq9_unpack(&out[i * 16 + 8]));
            if (!(gx == want.x) || !(gy == want.y)) bad++;
            checked++;
        }
        printf("check: %u of %u sampled sums differ from the XYZZ formulas' affine result%s\n", bad, checked, bad ? "  <-- WRONG" : " (OK)");
        if (bad) return 1;
    }
    // the product's arithmetic on the same gathers: a chain of K mixed XYZZ additions per lane (one 64-byte gather each)
    switch (K) {
    case 16: msx = run_chain<16, true>(d_tab, d_ia, d_out, n, reps); break;
    case 32: msx = run_chain<32, true>(d_tab, d_ia, d_out, n, reps); break;
    case 64: msx = run_chain<64, true>(d_tab, d_ia, d_out, n, reps); break;
    }
    printf("           XYZZ mixed-addition chains of %d on the same gathers: %.3f ms = %.2f G additions/s  (ratio %.2f)\n", K, msx,
           n / msx / 1e6, msx / ms);
    return 0;
}
