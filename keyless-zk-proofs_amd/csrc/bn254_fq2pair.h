// bn254_fq2pair.h -- Fq2 = Fq[u]/(u^2+1) spread over a LANE PAIR (device only), for the latency-bound G2 reductions.
//
// Why: the fold / weighted-sum stages of a G2 MSM (k_fold_all, k_fold_giant_b, k_wsum_level1, k_wsum_bits on Eng2n) are
// chains and trees of ~35 DEPENDENT XYZZ additions run by a few hundred lone wavefronts: what they cost is the time ONE
// addition takes ONE lane (14 Fq2 products = 6,800 multiply-adds, ~35 us at a lone wave's issue rate), and a lane that holds
// whole Fq2 values needs 308 registers -- one wave per SIMD, resident beside nothing (the prover's B2
// chain ends 0.7 ms after its G1 siblings and sits on the H MSM's sort).  Here the two components of every Fq2 value live on
// ADJACENT lanes: the even lane of a pair holds the real part a, the odd lane the imaginary part b, of every coordinate of the
// point the pair owns.  Additions, subtractions and partial reductions are component-wise (no exchange); a multiplication is ONE
// two-product reduction per lane (fmul9_sum2: 243 multiply-adds instead of 486), the partner's components crossing by DPP
// quad_perm moves (full-rate register moves, no LDS); the zero tests of the point formulas exchange one flag.  Half the
// dependent multiply-adds per addition, half the registers (<= 168: three waves per SIMD), twice the lanes.
//
// Same values as Fq2n (bn254_fq9.h: f2field.cpp:122-176), same invariant (every component < 2p + eps, normalised), so the
// generic XYZZ templates of bn254_curve.h (curve.cpp:91-458) instantiate on it unchanged; BOTH lanes of a pair always take the
// same branch (is_zero() combines the two components' tests).
#pragma once
#include "bn254_fq9.h"

namespace k16 {

// the partner lane's copy of a 32-bit value (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ uint32_t pair_xchg_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
__device__ __forceinline__ Fq9 pair_xchg(const Fq9& v)
{
    Fq9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = pair_xchg_u32(v.l[i]);
    return r;
}
__device__ __forceinline__ bool pair_is_odd() { return (threadIdx.x & 1u) != 0; } // workgroups are multiples of 64 lanes

struct Fq2h {
    Fq9 v; // even lane: a, odd lane: b
    static __device__ __forceinline__ Fq2h zero() { return Fq2h{fq9_zero()}; }
    static __device__ __forceinline__ Fq2h one() { return Fq2h{pair_is_odd() ? fq9_zero() : fq9_one()}; }
    __device__ __forceinline__ bool        is_zero() const
    {
        const uint32_t z = fq9_is_zero_mod_p<3>(v) ? 1u : 0u;
        return (z & pair_xchg_u32(z)) != 0;
    }
};
__device__ __forceinline__ Fq2h fadd(const Fq2h& x, const Fq2h& y) { return Fq2h{fred9(fadd9(x.v, y.v))}; }
__device__ __forceinline__ Fq2h fsub(const Fq2h& x, const Fq2h& y) { return Fq2h{fred9(fsub9<4>(x.v, y.v))}; }
__device__ __forceinline__ Fq2h fdbl(const Fq2h& x) { return Fq2h{fred9(fdbl9(x.v))}; }
__device__ __forceinline__ Fq2h fneg(const Fq2h& x) { return fsub(Fq2h::zero(), x); }
// (a + bu)(c + du) = (ac - bd) + (ad + bc)u.  With o = own component, p = the partner's:
//   even lane (o = a, c ; p = b, d):  a c + (4p - b) d  =  A o_y + B p_y   with A = o_x,  B = 4p - p_x (lazy limbs)
//   odd lane  (o = b, d ; p = a, c):  a d + b c         =  A o_y + B p_y   with A = p_x,  B = o_x
// bounds 2*2 + 4*2 = 12 of the 128 a reduction allows; column bound with the lazy operand 45 * 2^58 < 2^64.
__device__ __forceinline__ Fq2h fmul(const Fq2h& x, const Fq2h& y)
{
    const bool odd = pair_is_odd();
    const Fq9  px = pair_xchg(x.v), py = pair_xchg(y.v);
    const Fq9  npx = fsub9_lazy4_t<Fq9C>(fq9_zero(), px);
    Fq9        A, B;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        A.l[i] = odd ? px.l[i] : x.v.l[i];
        B.l[i] = odd ? x.v.l[i] : npx.l[i];
    }
    return Fq2h{fmul9_sum2(A, y.v, B, py)};
}
// (a + bu)^2 = (a + b)(a - b) + 2ab u: one product per lane -- even: (o + p)(o - p + 4p), odd: (2 p)(o)
__device__ __forceinline__ Fq2h fsqr(const Fq2h& x)
{
    const bool odd = pair_is_odd();
    const Fq9  p = pair_xchg(x.v);
    const Fq9  S = fadd9(x.v, p), D = fsub9<4>(x.v, p), P2 = fdbl9(p); // < 4p+, < 6p+, < 4p+
    Fq9        L, R;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        L.l[i] = odd ? P2.l[i] : S.l[i];
        R.l[i] = odd ? x.v.l[i] : D.l[i];
    }
    return Fq2h{fmul9(L, R)}; // 4 * 6 = 24 / 4 * 2 = 8 of 128
}

} // namespace k16
