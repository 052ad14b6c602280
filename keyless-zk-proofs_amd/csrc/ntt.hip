// ntt.hip -- radix-2 NTT / iNTT over BN254 Fr on the device.
//
// Replaces FFT<RawFr> (rust-rapidsnark/rapidsnark/src/fft.cpp): root table :40-136, bit-reversal
// :170-189, in-place DIT stages :192-219, inverse = forward + index reversal + 2^-k scale :222-246.
// Values are Montgomery-form Fr, 32 B little-endian, exactly as the reference holds them, and the
// outputs are bit-identical (same roots g^i with g = 5^((r-1)/2^S), same butterfly arithmetic;
// field results are canonical so evaluation order cannot change them).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "ctx.h"
#include "bn254_fq9.h"

#ifndef K16_CHAIN_PRIO
// Wave priority of the polynomial chain's kernels.  3 (above everything) until round 6; the witness MSMs' short kernels run at 3
// too and their accumulations at 0.  With the chain at 1 it still wins against the accumulations it runs beside, but the
// witness MSMs' fold / weighted-sum tails -- whose end, not the chain's, is what the H accumulation's start waits for -- are no
// longer held up by NTT waves: p50 over 11 alternating runs on two boxes 5.35-5.57 (median 5.46) against 5.37-5.84 (5.67) ms, equal on
// a third, two provers unchanged (profiles/r06/ab_wave_priorities.log, ab_chain_priority_second_box.log).  -DK16_CHAIN_PRIO=n to compare.
#define K16_CHAIN_PRIO 1
#endif
using namespace k16;

namespace {

__device__ __forceinline__ Fr ld_fr(const Fr* p)
{
    Fr           r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint4        a = s[0], b = s[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void st_fr(Fr* p, const Fr& r)
{
    uint4* d = reinterpret_cast<uint4*>(p);
    d[0]     = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    d[1]     = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// roots[i] = g^i from the table pw[k] = g^(2^k)   (fft.cpp:104-125 builds the same table serially)
struct PowTable {
    Fr pw[32];
};
__global__ void __launch_bounds__(256) k_build_roots(Fr* __restrict__ roots, uint32_t s, PowTable t)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << s)) return;
    Fr acc = Fr::one();
    for (uint32_t k = 0; k < s; k++)
        if ((i >> k) & 1) acc = fmul(acc, t.pw[k]);
    st_fr(&roots[i], acc);
}

// fft.cpp:170-189
__global__ void __launch_bounds__(256) k_bitrev(Fr* __restrict__ a, uint32_t logn)
{
    __builtin_amdgcn_s_setprio(K16_CHAIN_PRIO); // the polynomial chain gates the H MSM: its waves win VALU arbitration beside the witness MSMs
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << logn)) return;
    uint32_t r = __brev(i) >> (32 - logn);
    if (i > r) {
        Fr x = ld_fr(&a[i]), y = ld_fr(&a[r]);
        st_fr(&a[i], y);
        st_fr(&a[r], x);
    }
}

// fft.cpp:197-218 : one DIT stage, n/2 butterflies
__global__ void __launch_bounds__(256) k_stage(Fr* __restrict__ a, const Fr* __restrict__ roots, uint32_t logn,
                                               uint32_t s, uint32_t S)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << (logn - 1))) return;
    uint32_t md2 = 1u << (s - 1);
    uint32_t j   = i & (md2 - 1);
    uint32_t k   = (i >> (s - 1)) << s;
    Fr       w   = ld_fr(&roots[(size_t)j << (S - s)]);
    Fr       t   = fmul(w, ld_fr(&a[k + j + md2]));
    Fr       u   = ld_fr(&a[k + j]);
    st_fr(&a[k + j], fadd(t, u));
    st_fr(&a[k + j + md2], fsub(u, t));
}

// Fused radix-2^K passes (k_ntt_pass9 below): stages s0+1 .. s0+K of the same DIT network (fft.cpp:197-218)
// with the tile held in LDS, so the 21 stage launches of a 2^21 transform (21 x 128 MB of HBM traffic)
// become 3 (stages 1-10, 11-16, 17-21).  Element i = hi*2^(s0+K) + mid*2^s0 + lo: a workgroup owns one hi,
// T = 2^TL consecutive lo (coalesced 32*T-byte runs) and all 2^K mid; stage s0+t pairs mid with
// mid ^ 2^(t-1) and uses the twiddle roots[(mid_low*2^s0 + lo) << (S - s0 - t)] -- exactly the
// reference's root(s, j).  Same field values in the same dataflow => bit-identical results.

// fft.cpp:226-245 : a[i] <-> a[n-i], both scaled by 2^-logn; a[0], a[n/2] scaled in place
__global__ void __launch_bounds__(256) k_inv_tail(Fr* __restrict__ a, uint32_t logn, Fr scale)
{
    uint32_t n = 1u << logn;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > (n >> 1)) return;
    if (i == 0 || i == (n >> 1)) {
        if (i < n) st_fr(&a[i], fmul(ld_fr(&a[i]), scale));
        return;
    }
    Fr x = ld_fr(&a[i]), y = ld_fr(&a[n - i]);
    st_fr(&a[i], fmul(y, scale));
    st_fr(&a[n - i], fmul(x, scale));
}

// ---- the same pass on the radix-2^29 representation of Fr (bn254_fq9.h): butterflies are multiply-issue
// bound, and Fr9 multiplies 1.86x faster.  Data in HBM is "packed R'": x * 2^261 mod r (< 2r) in 32 bytes;
// CONV_IN / CONV_OUT convert from / to the reference's canonical Montgomery form at the ends of a
// transform (the public k16_ntt entry point); the prover keeps its whole a/b/c chain in the packed form.
__device__ __forceinline__ Fr9 ld_r9(const Fr* p)
{
    Fr w = ld_fr(p);
    return fr9_load(w.v);
}
__device__ __forceinline__ void st_r9(Fr* p, const Fr9& v)
{
    Fr w;
    fr9_store(w.v, v);
    st_fr(p, w);
}
// Twiddles are stored UNPACKED, nine 29-bit limbs (36 bytes) per root: a double stage loads three of them, and unpacking a
// 32-byte value costs 17 instructions each time (round 3; the data in HBM stays packed: it is loaded once per pass).
__device__ __forceinline__ Fr9 ld_tw9(const uint32_t* __restrict__ tw, size_t i)
{
    Fr9             r;
    const uint32_t* p = tw + i * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = p[k];
    return r;
}
// root(sp, j) -- the twiddle of butterfly j of stage sp (fft.hpp:40-43) -- from the stage-major copy of the late stages where
// there is one (k16_ntt_table::stage9: contiguous in j), from the root table otherwise (stride 2^(S - sp)).  The choice is
// uniform over a workgroup's double stage.
constexpr uint32_t STAGE9_LO = 17;
__device__ __forceinline__ Fr9 ld_stage_tw9(const uint32_t* __restrict__ roots9, const uint32_t* __restrict__ stage9, uint32_t S,
                                            uint32_t sp, size_t j)
{
#ifdef K16_LAB_NTT_TW0 // lab builds only (build_alt.sh NAME -DK16_LAB_NTT_TW0=<first stage> ntt): every twiddle of the stages from
    // that one on is root 0 -- WRONG transforms on purpose: what a pass costs when its twiddle loads all hit one cache line
    if (sp >= (K16_LAB_NTT_TW0)) j = 0;
#endif
    if (stage9 && sp >= STAGE9_LO && sp < S) return ld_tw9(stage9, ((size_t)1 << (sp - 1)) - ((size_t)1 << (STAGE9_LO - 1)) + j);
    return ld_tw9(roots9, j << (S - sp));
}
// stage9[2^(sp-1) - 2^16 + j] = roots9[j << (S - sp)] for STAGE9_LO <= sp < S, j < 2^(sp-1)
__global__ void __launch_bounds__(256) k_build_stage9(uint32_t* __restrict__ stage9, const uint32_t* __restrict__ roots9, uint32_t S,
                                                      uint64_t entries)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= entries) return;
    const uint64_t v  = e + ((uint64_t)1 << (STAGE9_LO - 1)); // in [2^(sp-1), 2^sp)
    const uint32_t sp = 64u - (uint32_t)__clzll((long long)v);
    const uint64_t j  = v - ((uint64_t)1 << (sp - 1));
    const uint32_t* src = roots9 + (j << (S - sp)) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) stage9[e * 9 + k] = src[k];
}
// Up to three polynomials per launch (blockIdx.y): the prover's a / b / c chains are independent, and a pass of one
// polynomial spends a quarter of its time filling and draining the chip (all resident workgroups load before any computes).
struct NttPtrs {
    Fr* src[3];
    Fr* dst[3]; // TAIL only
};
// TAIL (the last pass of the prover's inverse transforms): instead of storing the tile in place, apply what the
// reference does between the inverse and the coset-forward transform -- iNTT tail (fft.cpp:226-245: y[i] = X[(n-i) mod n] *
// 2^-k), coset shift (y[i] *= g^i, groth16.cpp:196-205) and the bit reversal that opens the forward transform (fft.cpp:170-189):
//     dst[bitrev(i)] = X[(n - i) mod n] * shift9[i],   shift9[i] = 2^-k * g^i  (one table, one multiplication)
// The tile is read mid-major for that store (the mid bits are the LOW bits of bitrev(i): runs of 2^K * 32 bytes in dst),
// with one element of padding per tile row so that the LDS reads stay conflict-free.
template <bool CONV_IN, bool CONV_OUT, bool TAIL, int THREADS = 256>
__global__ void __launch_bounds__(THREADS) k_ntt_pass9(NttPtrs pp, const uint32_t* __restrict__ roots9, uint32_t s0, uint32_t K,
                                                   uint32_t TL, uint32_t S, uint32_t logn, const Fr* __restrict__ shift9,
                                                   const uint32_t* __restrict__ stage9)
{
    __builtin_amdgcn_s_setprio(K16_CHAIN_PRIO); // the polynomial chain gates the H MSM: its waves win VALU arbitration beside the witness MSMs
    extern __shared__ uint4 ntt_lds[];
    uint32_t*      lds32 = reinterpret_cast<uint32_t*>(ntt_lds);
    // element e of the tile: nine dwords; the TAIL variant (which reads the tile mid-major: a stride of 9 T dwords between
    // adjacent lanes) skips one dword after every 64 elements, which spreads ANY power-of-two element stride over all 64
    // banks (round 3 padded every tile row by a whole element: 50 % more LDS at T = 2)
    auto at = [&](uint32_t e) -> Fr9& { return *reinterpret_cast<Fr9*>(lds32 + e * 9u + (TAIL ? (e >> 6) : 0u)); };
    Fr* __restrict__ a   = pp.src[blockIdx.y];
    const uint32_t T     = 1u << TL;
    const uint32_t RS    = T; // tile row stride in elements
    const uint32_t telem = T << K;
    const uint32_t lo_tiles = (1u << s0) >> TL;
    const uint32_t hi    = blockIdx.x / lo_tiles;
    const uint32_t lo0   = (blockIdx.x % lo_tiles) << TL;
    const size_t   base  = ((size_t)hi << (s0 + K)) + lo0;
    for (uint32_t e = threadIdx.x; e < telem; e += blockDim.x) {
        const uint32_t mid = e >> TL, tl = e & (T - 1);
        const Fr*      src = &a[base + ((size_t)mid << s0) + tl];
        at(mid * RS + tl) = CONV_IN ? fr9_from_fr(ld_fr(src)) : ld_r9(src);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // Lazy reduction inside the pass: tile values enter < 2r and grow by at most 6r per double stage (stage t: u + t < +2r,
    // u - t + 4r; stage t+1: + 2r again, with every t < 2r fresh from a multiplication; the twiddle-free first two stages of a
    // transform end < 8r; an odd leading single stage adds 2r), so after K <= 10 stages (ntt_passes) they are < 32r < 2^259 --
    // within the 9 x 29-bit limbs, within the multiply's operand bound (2 * 32 <= 128) and within fred9's range; one fred9
    // at the store brings them back.
    const uint32_t nbf = telem >> 1;
    uint32_t       t   = 1;
    if (K & 1) { // odd stage count: one plain radix-2 stage first
        const uint32_t half = 1u;
        for (uint32_t b = threadIdx.x; b < nbf; b += blockDim.x) {
            const uint32_t tl = b & (T - 1), mm = b >> TL;
            const uint32_t ml = mm & (half - 1), mh = mm >> (t - 1);
            const uint32_t m0 = (mh << t) + ml, m1 = m0 + half;
            const size_t   j  = ((size_t)ml << s0) + lo0 + tl;
            Fr9            x1 = at(m1 * RS + tl);
            Fr9            u  = at(m0 * RS + tl);
            Fr9            tt = (s0 == 0) ? x1 : frmul9(ld_stage_tw9(roots9, stage9, S, s0 + t, j), x1); // s0 == 0: the twiddle is 1
            at(m0 * RS + tl) = fadd9(u, tt);
            at(m1 * RS + tl) = fsub9_t<Fr9C, 2>(u, tt);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        t = 2;
    }
    // two stages (t, t+1) per LDS round trip: a thread owns the 4 elements m0 + {0, h, 2h, 3h} (h = 2^(t-1)); stage t pairs
    // (m0, m0+h) and (m0+2h, m0+3h) with the SAME twiddle, stage t+1 pairs (m0, m0+2h) and (m0+h, m0+3h) -- exactly the
    // butterflies and twiddles of the two radix-2 stages (fft.cpp:197-218), with half the LDS traffic and barriers.
    const uint32_t nq = telem >> 2;
    for (; t + 1 <= K; t += 2) {
        const uint32_t h = 1u << (t - 1);
        for (uint32_t g = threadIdx.x; g < nq; g += blockDim.x) {
            const uint32_t tl = g & (T - 1), mm = g >> TL;
            const uint32_t ml = mm & (h - 1), mh = mm >> (t - 1);
            const uint32_t m0 = (mh << (t + 1)) + ml;
            const uint32_t i0 = m0 * RS + tl, dh = h * RS;
            const size_t   ja = ((size_t)ml << s0) + lo0 + tl;         // twiddle index of stage t, and of (m0, m0+2h) in t+1
            const size_t   jb = ((size_t)(ml + h) << s0) + lo0 + tl;   // (m0+h, m0+3h) in stage t+1
            Fr9            x0 = at(i0), x1 = at(i0 + dh), x2 = at(i0 + 2 * dh), x3 = at(i0 + 3 * dh);
            const bool     unit = (s0 == 0 && t == 1); // first two stages of a transform: ja = 0, the twiddles are 1
            Fr9            p1, p3;
            if (unit) {
                p1 = x1;
                p3 = x3;
            } else {
                Fr9 w1 = ld_stage_tw9(roots9, stage9, S, s0 + t, ja);
                p1     = frmul9(w1, x1);
                p3     = frmul9(w1, x3);
            }
            // Stage t's sums and differences are consumed once each -- a2 / a3 by a multiplication, a0 / a1 by a normalising
            // addition / subtraction -- so (outside the twiddle-free opening stages) they keep lazy limbs: 54 instead of 122
            // instructions for the four of them.  The differences then carry +4r instead of +2r (see fsub9_lazy4_t).
            Fr9 a0, a1, a2, a3;
            if (unit) {
                a0 = fadd9(x0, p1);
                a1 = fsub9_t<Fr9C, 2>(x0, p1);
                a2 = fadd9(x2, p3);
                a3 = fsub9_t<Fr9C, 2>(x2, p3);
            } else {
                a0 = fadd9_lazy(x0, p1);
                a1 = fsub9_lazy4_t<Fr9C>(x0, p1);
                a2 = fadd9_lazy(x2, p3);
                a3 = fsub9_lazy4_t<Fr9C>(x2, p3);
            }
            Fr9 q2 = unit ? a2 : frmul9(ld_stage_tw9(roots9, stage9, S, s0 + t + 1, ja), a2);
            Fr9 q3 = frmul9(ld_stage_tw9(roots9, stage9, S, s0 + t + 1, jb), a3);
            at(i0)          = fadd9(a0, q2);
            // unit: q2 = a2 = x2 + x3 is not fresh from a multiplication -- up to 4r, so the offset must be 4r (with 2r the
            // difference goes negative when x2, x3 >= r and x0 + x1 is small: about once per 10^3 proofs of 2^21 with
            // inputs < 1.006 r, every proof with inputs < 1.07 r)
            at(i0 + 2 * dh) = unit ? fsub9_t<Fr9C, 4>(a0, q2) : fsub9_t<Fr9C, 2>(a0, q2);
            at(i0 + dh)     = fadd9(a1, q3);
            at(i0 + 3 * dh) = fsub9_t<Fr9C, 2>(a1, q3);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (TAIL) {
        Fr* __restrict__ d    = pp.dst[blockIdx.y];
        const uint32_t   nm1  = (1u << logn) - 1;
        for (uint32_t e = threadIdx.x; e < telem; e += blockDim.x) {
            const uint32_t mid = e & ((1u << K) - 1), tl = e >> K;
            const uint32_t f   = (uint32_t)base + (mid << s0) + tl;
            const uint32_t i   = (0u - f) & nm1;                       // (n - f) mod n
            const uint32_t to  = logn ? (__brev(i) >> (32 - logn)) : 0u;
            st_r9(&d[to], frmul9(at(mid * RS + tl), ld_r9(&shift9[i]))); // 32 * 2 <= 128: < 2r
        }
        return;
    }
    for (uint32_t e = threadIdx.x; e < telem; e += blockDim.x) {
        const uint32_t mid = e >> TL, tl = e & (T - 1);
        Fr*            dst = &a[base + ((size_t)mid << s0) + tl];
        if (CONV_OUT)
            st_fr(dst, fr9_to_fr(at(mid * RS + tl)));             // multiply by 2^256 / 2^261: any bound <= 64r is fine
        else
            st_r9(dst, fred9_t<Fr9C>(at(mid * RS + tl)));         // < r (1 + 2^-17): fits the 32-byte packed form
    }
}
// shift9[i] = 2^-logn * g^i, g the primitive 2^(logn+1)-th root: the factor between the inverse and the coset-forward transform
__global__ void __launch_bounds__(256) k_build_shift9(Fr* __restrict__ shift9, const uint32_t* __restrict__ roots9, uint32_t n,
                                                      uint32_t stride_log, Fr9 scale)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st_r9(&shift9[i], fred9_t<Fr9C>(frmul9(ld_tw9(roots9, (size_t)i << stride_log), scale)));
}
// fft.cpp:226-245 on packed R' data
__global__ void __launch_bounds__(256) k_inv_tail9(Fr* __restrict__ a, uint32_t logn, Fr9 scale)
{
    __builtin_amdgcn_s_setprio(K16_CHAIN_PRIO); // the polynomial chain gates the H MSM: its waves win VALU arbitration beside the witness MSMs
    uint32_t n = 1u << logn;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > (n >> 1)) return;
    if (i == 0 || i == (n >> 1)) {
        if (i < n) st_r9(&a[i], frmul9(ld_r9(&a[i]), scale));
        return;
    }
    Fr9 x = ld_r9(&a[i]), y = ld_r9(&a[n - i]);
    st_r9(&a[i], frmul9(y, scale));
    st_r9(&a[n - i], frmul9(x, scale));
}
__global__ void __launch_bounds__(256) k_roots_to_r9(const Fr* __restrict__ roots, uint32_t* __restrict__ roots9, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr9 v = fr9_from_fr(ld_fr(&roots[i])); // < 2r, normalised
#pragma unroll
    for (int k = 0; k < 9; k++) roots9[i * 9 + k] = v.l[k];
}

uint32_t ilog2_u64(uint64_t n)
{
    uint32_t r = 0;
    while (n > 1) {
        n >>= 1;
        r++;
    }
    return r;
}

} // namespace

// fft.cpp:40-136
int k16_ntt_get_table(k16_ctx* ctx, uint64_t max_domain, k16_ntt_table** out)
{
    if (max_domain == 0 || (max_domain & (max_domain - 1))) {
        ctx->err = "ntt: domain must be a power of two";
        return K16_ERR_ARG;
    }
    uint32_t dp = ilog2_u64(max_domain);
    if (dp > 28) { // 2-adicity of r (fft.cpp:81-84 "Domain size too big for the curve")
        ctx->err = "ntt: domain size too big for the curve";
        return K16_ERR_ARG;
    }
    uint32_t s  = dp < 1 ? 1 : dp; // the reference's table always has s >= 1
    auto     it = ctx->ntt_tables.find(s);
    if (it != ctx->ntt_tables.end()) {
        *out = &it->second;
        return K16_OK;
    }
    k16_ntt_table t;
    t.s = s;
    // g = 5^((r-1)/2^s)   (nqr = 5 for BN254 r: fft.cpp:60-67)
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrParams::P[i];
    e[0] -= 1; // r - 1 (r is odd, no borrow)
    for (uint32_t k = 0; k < s; k++) {
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? e[i + 1] << 31 : 0);
    }
    Fr five = Fr::zero();
    five.v[0] = 5;
    five      = to_mont(five);
    Fr       g = fpow(five, e);
    PowTable pt;
    pt.pw[0] = g;
    for (int k = 1; k < 32; k++) pt.pw[k] = fsqr(pt.pw[k - 1]);
    // powTwoInv[k] = 2^-k (fft.cpp:99-102, 127-130)
    Fr two = Fr::zero();
    two.v[0] = 2;
    Fr half       = finv(to_mont(two));
    t.pow2inv[0]  = Fr::one();
    for (int k = 1; k <= 33; k++) t.pow2inv[k] = fmul(t.pow2inv[k - 1], half);
    K16_HIP(ctx, hipMalloc((void**)&t.roots, sizeof(Fr) << s));
    uint64_t nr = 1ull << s;
    hipLaunchKernelGGL(k_build_roots, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, ctx->stream, t.roots, s, pt);
    K16_HIP(ctx, hipGetLastError());
    K16_HIP(ctx, hipMalloc((void**)&t.roots9, (size_t)36 << s));
    hipLaunchKernelGGL(k_roots_to_r9, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, ctx->stream, t.roots, t.roots9, nr);
    K16_HIP(ctx, hipGetLastError());
    for (int k = 0; k <= 33; k++) t.pow2inv9[k] = fr9_from_fr(t.pow2inv[k]);
    if (s > STAGE9_LO) { // stage-major twiddles of stages 17 .. s - 1 (see k16_ntt_table::stage9): 73 MB for the prover's 2^22 table
        const uint64_t entries = ((uint64_t)1 << (s - 1)) - ((uint64_t)1 << (STAGE9_LO - 1));
        if (hipMalloc((void**)&t.stage9, (size_t)entries * 36) == hipSuccess) {
            hipLaunchKernelGGL(k_build_stage9, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, ctx->stream, t.stage9, t.roots9, s, entries);
            K16_HIP(ctx, hipGetLastError());
        } else {
            (void)hipGetLastError(); // no room: the passes read the root table (same values)
            t.stage9 = nullptr;
        }
    }
    auto ins = ctx->ntt_tables.emplace(s, t);
    *out     = &ins.first->second;
    return K16_OK;
}

// the fused passes of one transform over `count` polynomials; tail_dst != nullptr: the last pass stores through the TAIL path
static void ntt_passes(k16_ctx* ctx, Fr* const* polys, int count, uint32_t logn, k16_ntt_table* tab, bool packed9,
                       Fr* const* tail_dst, const Fr* shift9, hipStream_t st, uint32_t tile_log = 10)
{
    NttPtrs pp = {};
    for (int i = 0; i < count; i++) {
        pp.src[i] = polys[i];
        pp.dst[i] = tail_dst ? tail_dst[i] : nullptr;
    }
    const uint64_t  n      = 1ull << logn;
    uint32_t        s0     = 0;
    const uint32_t* stage9 = ctx->tune.ntt_no_stage_tables ? nullptr : tab->stage9;
    // Tile size.  1024 elements (36 KB of LDS, four workgroups per CU): 21 stages are three passes (10 + 6 + 5, 512-byte runs).
    // K16_NTT_TILE_LOG=11: 2048 elements (72 KB, two workgroups of 512 threads per CU), two passes (11 + 10) -- one load /
    // store round trip less per transform, but the second pass then moves 64-byte runs (T = 2: a tile is T x 2^10 elements
    // whatever the layout between the passes).  Measured in a proof (three polynomials per launch, 2^21, round 4): forward
    // 395 + 684 us against 610 + 349 + 300, the inverse with its TAIL store 372 + 1194 against 309 + 393 + 450; proof p50
    // 6.03-6.18 (both), 5.85-6.06 (forward only) against 5.86-6.05 ms: no gain, so 1024 stays the default.
    if (tail_dst && ctx->tune.ntt_tail_small) tile_log = 10;
    const uint32_t TLMAX = tile_log == 11 ? 1u : 4u;
    while (s0 < logn) {
        uint32_t       TL = s0 < TLMAX ? s0 : TLMAX;                    // T = min(2^s0, 16) lo values per tile
        const uint32_t K  = std::min<uint32_t>(logn - s0, tile_log - TL); // <= 2^tile_log elements per tile
        // a SHORT last pass (2^21: stages 17-21, K = 5) fills its tile with more lo values instead of running half-empty: with
        // T = 16 its tiles had 512 elements -- 128 quads for 256 lanes, so the two double stages of its three rounds ran on
        // half of the workgroup (the pass sat at 0.60 of its issue bound, profiles/r04/pmc_ntt_passes.txt); T = 32 gives every lane
        // its four elements and 1 KB runs (round 5)
        if (TL + K < tile_log && !ctx->tune.ntt_tail_small) TL = std::min<uint32_t>(s0, tile_log - K);
        const bool     first = s0 == 0, last = s0 + K == logn;
        const bool     cin = !packed9 && first, cout = !packed9 && last, tail = tail_dst && last;
        const dim3     grid((unsigned)(n >> (K + TL)), (unsigned)count);
        const size_t   telem = (size_t)1 << (TL + K);
        const size_t   lds   = telem * sizeof(Fr9) + (tail ? (telem >> 6) * 4 + 16 : 0);
        const bool     big = lds > 48 * 1024; // 512 threads per workgroup, dynamic LDS above the default limit
        // K16_OPT_SHARED_GPU / K16_NTT_WG_PER_CU=3: the 1024-element passes ask for a third of the CU's LDS instead of the 36 KB
        // they use, so that three workgroups are resident per CU instead of four -- 312 of a SIMD's 512 registers, which leaves
        // room for a wave of a bucket accumulation (159) beside them
        const unsigned wg_per_cu = ctx->ntt_wg_per_cu;
        const size_t lds_launch = (!big && wg_per_cu < 4) ? std::max<size_t>(lds, (size_t)(160 * 1024 / wg_per_cu) - 1024) : lds;
#define K16_NTT_LAUNCH(CI, CO, TA)                                                                                              \
    do {                                                                                                                        \
        if (!big && lds_launch > lds) {                                                                                         \
            static bool attr2 = false;                                                                                          \
            if (!attr2) {                                                                                                       \
                (void)hipFuncSetAttribute((const void*)k_ntt_pass9<CI, CO, TA, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          160 * 1024);                                                                          \
                attr2 = true;                                                                                                   \
            }                                                                                                                   \
            hipLaunchKernelGGL((k_ntt_pass9<CI, CO, TA, 256>), grid, dim3(256), lds_launch, st, pp, tab->roots9, s0, K, TL,     \
                               tab->s, logn, shift9, stage9);                                                                   \
        } else if (big) {                                                                                                              \
            static bool attr = false;                                                                                           \
            if (!attr) {                                                                                                        \
                (void)hipFuncSetAttribute((const void*)k_ntt_pass9<CI, CO, TA, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          128 * 1024);                                                                          \
                attr = true;                                                                                                    \
            }                                                                                                                   \
            hipLaunchKernelGGL((k_ntt_pass9<CI, CO, TA, 512>), grid, dim3(512), lds, st, pp, tab->roots9, s0, K, TL, tab->s,    \
                               logn, shift9, stage9);                                                                           \
        } else                                                                                                                  \
            hipLaunchKernelGGL((k_ntt_pass9<CI, CO, TA, 256>), grid, dim3(256), lds, st, pp, tab->roots9, s0, K, TL, tab->s,    \
                               logn, shift9, stage9);                                                                           \
    } while (0)
        if (tail)
            K16_NTT_LAUNCH(false, false, true);
        else if (cin && cout)
            K16_NTT_LAUNCH(true, true, false);
        else if (cin)
            K16_NTT_LAUNCH(true, false, false);
        else if (cout)
            K16_NTT_LAUNCH(false, true, false);
        else
            K16_NTT_LAUNCH(false, false, false);
#undef K16_NTT_LAUNCH
        s0 += K;
    }
}

int k16_ntt_enqueue(k16_ctx* ctx, k16::Fr* d_a, uint64_t n, k16_ntt_table* tab, int inverse, hipStream_t st, int packed9)
{
    if (!st) st = ctx->stream;
    if (n == 0 || (n & (n - 1)) || n > (1ull << tab->s)) {
        ctx->err = "ntt: n must be a power of two <= table size";
        return K16_ERR_ARG;
    }
    uint32_t       logn = ilog2_u64(n);
    k16_stat_scope ss(ctx, "ntt", st);
    const bool skip_bitrev = (packed9 & 2) != 0, skip_tail = (packed9 & 4) != 0;
    packed9 &= 1;
    if (logn >= 1) {
        if (!skip_bitrev) hipLaunchKernelGGL(k_bitrev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_a, logn);
        if (!packed9 && ctx->tune.ntt_unfused) {
            for (uint32_t s = 1; s <= logn; s++)
                hipLaunchKernelGGL(k_stage, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, d_a, tab->roots, logn,
                                   s, tab->s);
        } else {
            Fr* one[1] = {d_a};
            const uint32_t pub_tile = ctx->tune.ntt_tile_log ? (uint32_t)std::max(10, std::min(11, ctx->tune.ntt_tile_log)) : 10u;
            ntt_passes(ctx, one, 1, logn, tab, packed9 != 0, nullptr, nullptr, st, logn >= 12 ? pub_tile : 10u);
        }
    }
    if (inverse && !skip_tail) {
        if (logn == 0) {
            // n == 1: fft.cpp:243-244 scales a[0] twice (a[0] and a[n>>1] alias) by 2^0 = 1: identity
        } else if (packed9) {
            hipLaunchKernelGGL(k_inv_tail9, dim3((unsigned)((n / 2 + 1 + 255) / 256)), dim3(256), 0, st, d_a, logn,
                               tab->pow2inv9[logn]);
        } else {
            hipLaunchKernelGGL(k_inv_tail, dim3((unsigned)((n / 2 + 1 + 255) / 256)), dim3(256), 0, st, d_a,
                               logn, tab->pow2inv[logn]);
        }
    }
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}

// The prover's coset chain on `count` <= 3 polynomials at once (groth16.cpp:172-262: ifft, coset shift, fft per polynomial):
// src[k] holds packed R' data in bit-reversed order (the SpMV writes it that way); the inverse passes run in place, their
// last pass stores [tail, shift, bit reversal] into dst[k] (k_ntt_pass9<.., TAIL>), the forward passes run in place on dst[k].
// shift9 comes from k16_ntt_build_coset_shift for this n.  src and dst must not overlap.
int k16_ntt_coset_chain(k16_ctx* ctx, k16::Fr* const* src, k16::Fr* const* dst, int count, uint64_t n, k16_ntt_table* tab,
                        const k16::Fr* shift9, hipStream_t st)
{
    if (count < 1 || count > 3 || n < 1 || (n & (n - 1)) || 2 * n > (1ull << tab->s)) {
        ctx->err = "ntt: coset chain needs 1-3 polynomials of a power-of-two size within the table";
        return K16_ERR_ARG;
    }
    const uint32_t logn = ilog2_u64(n);
    if (logn == 0) { // one point: every step of the chain is the identity
        for (int k = 0; k < count; k++) K16_HIP(ctx, hipMemcpyAsync(dst[k], src[k], sizeof(Fr), hipMemcpyDeviceToDevice, st));
        return K16_OK;
    }
    k16_stat_scope ss(ctx, "ntt", st);
    const uint32_t fwd_tile = ctx->tune.ntt_tile_log ? (uint32_t)std::max(10, std::min(11, ctx->tune.ntt_tile_log)) : 10u;
    ntt_passes(ctx, src, count, logn, tab, true, dst, shift9, st, logn >= 12 ? fwd_tile : 10u);
    ntt_passes(ctx, dst, count, logn, tab, true, nullptr, nullptr, st, logn >= 12 ? fwd_tile : 10u);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}

int k16_ntt_build_coset_shift(k16_ctx* ctx, k16_ntt_table* tab, uint64_t n, k16::Fr** out, hipStream_t st)
{
    const uint32_t logn = ilog2_u64(n);
    if (n < 1 || (n & (n - 1)) || logn + 1 > tab->s) {
        ctx->err = "ntt: coset shift table needs the roots of twice the domain";
        return K16_ERR_ARG;
    }
    K16_HIP(ctx, hipMalloc((void**)out, (size_t)n * sizeof(Fr)));
    hipLaunchKernelGGL(k_build_shift9, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st ? st : ctx->stream, *out, tab->roots9,
                       (uint32_t)n, tab->s - logn - 1, tab->pow2inv9[logn]);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}

extern "C" int k16_ntt(k16_ctx* ctx, void* d_a, uint64_t n, uint64_t max_domain, int inverse)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !d_a) return K16_ERR_ARG;
    k16_ntt_table* tab = nullptr;
    int            rc  = k16_ntt_get_table(ctx, max_domain, &tab);
    if (rc) return rc;
    return k16_ntt_enqueue(ctx, (Fr*)d_a, n, tab, inverse, nullptr, 0);
    });
}

extern "C" int k16_ntt_host(k16_ctx* ctx, void* h_a, uint64_t n, uint64_t max_domain, int inverse)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !h_a) return K16_ERR_ARG;
    void* d = nullptr;
    K16_HIP(ctx, hipMalloc(&d, n * 32));
    int rc = K16_OK;
    if (hipMemcpyAsync(d, h_a, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = K16_ERR_HIP;
    if (!rc) rc = k16_ntt(ctx, d, n, max_domain, inverse);
    if (!rc && hipMemcpyAsync(h_a, d, n * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = K16_ERR_HIP;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
    });
}
