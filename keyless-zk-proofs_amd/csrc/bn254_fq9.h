// bn254_fq9.h -- BN254 Fq in an unsaturated radix-2^29 representation (9 limbs), for the EC hot loops.
//
// Why: gfx950's integer multiplier is v_mad_u64_u32 (32x32+64 -> 64, half rate) and every carry
// (v_add_co / v_addc) costs about as much as a multiply-add (profiles/r01/ubench_instruction_rates.log).
// With 29-bit limbs a 64-bit accumulator can absorb all 18 partial products of a column
// (18 * 2^58 < 2^63), so a Montgomery multiplication is 162 carry-free multiply-adds plus one
// shift/mask per column -- 1.9x the throughput of the canonical 8 x 32-bit CIOS (bn254_field.h).
//
// Semantics.  An Fq9 holds an integer V = sum l[i] * 2^(29 i) with l[0..7] < 2^29 ("normalised"),
// congruent mod p to x * R' where R' = 2^261 (Montgomery form for THIS radix).  V is NOT kept
// canonical: each function documents the bound (as a multiple of p) it needs and returns; p < 2^254
// and R'/p ~ 169, so  fmul9(a, b) < p * (1 + A*B/169)  for a < A*p, b < B*p  -- i.e. < 2p whenever
// A*B <= 128, with no final conditional subtraction at all.  Additions and subtractions just add
// limbs (plus a multiple of p for subtraction) and renormalise.  Conversions to and from the
// reference's canonical Montgomery form (R = 2^256, fq_raw_generic.cpp) are exact, so the group
// elements computed here are the same as with bn254_field.h; only the schedule of reductions differs.
#pragma once
#include "bn254_field.h"

namespace k16 {

struct Fq9 {
    uint32_t l[9];
};

struct Fq9C {
static constexpr uint32_t MASK = (1u << 29) - 1;
// p in radix 2^29
static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                           0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
static constexpr uint32_t NP   = 0x04866389u; // -p^-1 mod 2^29
// R' mod p  (Montgomery one)
static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                             0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
// 2^266 mod p = R'^2 / R : fmul9(x*R, K_IN) = x*R'
static constexpr uint32_t K_IN[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                              0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
// 2^256 mod p = R : fmul9(x*R', K_OUT) = x*R
static constexpr uint32_t K_OUT[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u,
                               0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
// k*p for the subtraction offsets
static constexpr uint32_t KP2[9] = {0x10f9fa8eu, 0x0208c16du, 0x18e5469eu, 0x05aa45a1u, 0x0b0bb2f0u,
                             0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
static constexpr uint32_t KP4[9] = {0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u,
                             0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
static constexpr uint32_t KP8[9] = {0x03e7ea38u, 0x082305b6u, 0x03951a78u, 0x16a91687u, 0x0c2ecbc0u,
                             0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
// 3p, 6p, 10p: offsets of the bucket accumulator's lazy subtractions (acc9_madd)
static constexpr uint32_t KP3[9] = {0x0976f7d5u, 0x030d2224u, 0x1557e9edu, 0x087f6872u, 0x00918c68u,
                             0x0891c242u, 0x01f4a3c3u, 0x0b14729cu, 0x00912cebu};
static constexpr uint32_t KP6[9] = {0x12edefaau, 0x061a4448u, 0x0aafd3dau, 0x10fed0e5u, 0x012318d0u,
                             0x11238484u, 0x03e94786u, 0x1628e538u, 0x012259d6u};
static constexpr uint32_t KP10[9] = {0x14e1e4c6u, 0x0a2bc723u, 0x1c7a6116u, 0x1c535c28u, 0x173a7eb0u,
                              0x1c908786u, 0x0684cc8au, 0x0f997e08u, 0x01e3eb10u};
static constexpr uint32_t PINV = 0x1b799c77u; // p^-1 mod 2^29
};
// the same for the scalar field r (NTT chain): r in radix 2^29, -r^-1 mod 2^29, 2^261 / 2^266 / 2^256 mod r, k*r
struct Fr9C {
    static constexpr uint32_t MASK = (1u << 29) - 1;
    static constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NP   = 0x0fffffffu;
    static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t K_IN[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau,
                                         0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    static constexpr uint32_t K_OUT[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu,
                                          0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    static constexpr uint32_t KP2[9] = {0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u,
                                        0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
    static constexpr uint32_t KP4[9] = {0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu,
                                        0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
    static constexpr uint32_t KP8[9] = {0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u,
                                        0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
};


K16_HD Fq9 fq9_zero()
{
    Fq9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
K16_HD Fq9 fq9_one()
{
    Fq9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = Fq9C::ONE[i];
    return r;
}
K16_HD bool fq9_limbs_zero(const Fq9& a)
{
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}

// K16_FMUL_CHAINS: accumulator chains per column of a product (2: the reduction terms run on a chain of their own and
// are merged once per column -- half the dependent multiply-add depth, 17 more 64-bit additions per product; 1: one chain).
#ifndef K16_FMUL_CHAINS
#define K16_FMUL_CHAINS 2
#endif

// Montgomery product for radix 2^29: returns a*b/R' mod p, < p*(1 + A*B/169); limbs normalised.
// Needs normalised inputs (l[0..7] < 2^29, l[8] < 2^29).  Product scanning; column k collects the
// a_i*b_j with i+j = k and the m_i*p_j reduction terms; m_k clears the low 29 bits of column k.
template <class C>
K16_HD Fq9 fmul9_t(const Fq9& a, const Fq9& b)
{
    uint32_t m[9];
    Fq9      r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t  acc2 = 0; // second chain: halves the dependent-multiply-add depth per column
        uint64_t& red  = K16_FMUL_CHAINS == 1 ? acc : acc2;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a.l[i] * b.l[j];
            if (k >= 9 || i < k) red += (uint64_t)m[i] * C::P[j];
        }
        if (K16_FMUL_CHAINS != 1) acc += acc2;
        if (k < 9) {
            m[k] = ((uint32_t)acc * C::NP) & C::MASK;
            acc += (uint64_t)m[k] * C::P[0];
            acc >>= 29;
        } else {
            r.l[k - 9] = (uint32_t)acc & C::MASK;
            acc >>= 29;
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
K16_HD Fq9 fmul9(const Fq9& a, const Fq9& b) { return fmul9_t<Fq9C>(a, b); }

// a * v / R' for a single-limb v < 2^29: fmul9_t with b = {v, 0, ..., 0} -- the same integer product, hence the same limbs,
// for 9 + 81 multiply-adds instead of 162 (the SpMV's witness values are below 256 for 98 % of the wires).
template <class C>
K16_HD Fq9 fmul9_small_t(const Fq9& a, uint32_t v)
{
    uint32_t m[9];
    Fq9      r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc2 = 0;
        if (k < 9) acc += (uint64_t)a.l[k] * v;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            if (k >= 9 || i < k) acc2 += (uint64_t)m[i] * C::P[j];
        }
        acc += acc2;
        if (k < 9) {
            m[k] = ((uint32_t)acc * C::NP) & C::MASK;
            acc += (uint64_t)m[k] * C::P[0];
            acc >>= 29;
        } else {
            r.l[k - 9] = (uint32_t)acc & C::MASK;
            acc >>= 29;
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// a^2 / R': the 36 off-diagonal products are formed once against the doubled limbs (2*a_i < 2^30), so a
// squaring is 45 + 81 multiply-adds instead of 162.  Column bound: 4 * 2^59 + 2^58 + 9 * 2^58 < 2^63.
template <class C>
K16_HD Fq9 fsqr9_t(const Fq9& a)
{
    uint32_t m[9], a2[9];
    Fq9      r;
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t  acc2 = 0;
        uint64_t& red  = K16_FMUL_CHAINS == 1 ? acc : acc2;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            if (i < j) acc += (uint64_t)a2[i] * a.l[j];
            if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
            if (k >= 9 || i < k) red += (uint64_t)m[i] * C::P[j];
        }
        if (K16_FMUL_CHAINS != 1) acc += acc2;
        if (k < 9) {
            m[k] = ((uint32_t)acc * C::NP) & C::MASK;
            acc += (uint64_t)m[k] * C::P[0];
            acc >>= 29;
        } else {
            r.l[k - 9] = (uint32_t)acc & C::MASK;
            acc >>= 29;
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
K16_HD Fq9 fsqr9(const Fq9& a) { return fsqr9_t<Fq9C>(a); }

// (a*b + c*d) / R' with ONE Montgomery reduction: < p * (1 + (A*B + C*D)/169).  Column bound: 27 * 2^58 < 2^63.
template <class C>
K16_HD Fq9 fmul9_sum2_t(const Fq9& a, const Fq9& b, const Fq9& c, const Fq9& d)
{
    uint32_t m[9];
    Fq9      r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t  acc2 = 0, acc3 = 0;
        uint64_t& red  = K16_FMUL_CHAINS == 1 ? acc3 : acc2; // one chain less: the reduction terms ride on the second product's
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a.l[i] * b.l[j];
            acc3 += (uint64_t)c.l[i] * d.l[j];
            if (k >= 9 || i < k) red += (uint64_t)m[i] * C::P[j];
        }
        if (K16_FMUL_CHAINS != 1) acc += acc2 + acc3; else acc += acc3;
        if (k < 9) {
            m[k] = ((uint32_t)acc * C::NP) & C::MASK;
            acc += (uint64_t)m[k] * C::P[0];
            acc >>= 29;
        } else {
            r.l[k - 9] = (uint32_t)acc & C::MASK;
            acc >>= 29;
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
K16_HD Fq9 fmul9_sum2(const Fq9& a, const Fq9& b, const Fq9& c, const Fq9& d) { return fmul9_sum2_t<Fq9C>(a, b, c, d); }

// a + b (bound A + B); limbs renormalised
K16_HD Fq9 fadd9(const Fq9& a, const Fq9& b)
{
    Fq9      r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.l[i] + b.l[i] + c;
        r.l[i]     = t & Fq9C::MASK;
        c          = t >> 29;
    }
    r.l[8] = a.l[8] + b.l[8] + c;
    return r;
}
K16_HD Fq9 fdbl9(const Fq9& a) { return fadd9(a, a); }

// a - b + K*p, K in {2, 4, 8}; needs b < K*p; result < A + K.  Signed limb-wise difference with an
// arithmetic-shift carry; the total is non-negative so the top limb ends >= 0.
template <class C, int K>
K16_HD Fq9 fsub9_t(const Fq9& a, const Fq9& b)
{
    Fq9     r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t kp = K == 2 ? C::KP2[i] : (K == 4 ? C::KP4[i] : C::KP8[i]);
        int32_t        t  = (int32_t)(a.l[i] + kp) - (int32_t)b.l[i] + c;
        if (i < 8) {
            r.l[i] = (uint32_t)t & C::MASK;
            c      = t >> 29;
        } else {
            r.l[8] = (uint32_t)t;
        }
    }
    return r;
}

template <int K>
K16_HD Fq9 fsub9(const Fq9& a, const Fq9& b)
{
    return fsub9_t<Fq9C, K>(a, b);
}

// ---- the same without carry propagation ("lazy" limbs), for values that go straight into ONE normalising operation or
// into a multiplication as its only lazy operand (NTT butterflies, ntt.hip).
// a + b limb by limb: limbs < 2^30 for normalised inputs.  A multiplication takes such an operand against a normalised one:
// a column is at most 9 * 2^30 * 2^29 + 9 * 2^58 + carry < 2^64.
K16_HD Fq9 fadd9_lazy(const Fq9& a, const Fq9& b)
{
    Fq9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + 4p limb by limb.  4p is written with 2^29 lent to each of the limbs 0..7 by the limb above (l'[0] = l[0] + 2^29,
// l'[i] = l[i] + 2^29 - 1, l'[8] = l[8] - 1), so that no limb of the result is negative: needs b NORMALISED and b < 2p
// (top limb of b <= top limb of 2p < l[8](4p) - 1); a normalised.  Result limbs < 3 * 2^29 (a multiplication column against
// a normalised operand: 27 * 2^58 + 9 * 2^58 + carry < 2^64), value < A + 4.
template <class C>
K16_HD Fq9 fsub9_lazy4_t(const Fq9& a, const Fq9& b)
{
    Fq9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t kp = i == 0 ? C::KP4[0] + (1u << 29) : (i < 8 ? C::KP4[i] + (1u << 29) - 1u : C::KP4[8] - 1u);
        r.l[i]            = a.l[i] + kp - b.l[i];
    }
    return r;
}

// exact comparison with j*p, j = 0 .. J-1, for a normalised value < J*p:  V == 0 (mod p) ?
// Prefilter on the low limb (false positives ~ J / 2^29), then a full 9-limb compare.
template <int J>
K16_HD bool fq9_maybe_zero_mod_p(const Fq9& a)
{
    return ((a.l[0] * Fq9C::PINV) & Fq9C::MASK) < (uint32_t)J;
}
template <int J>
K16_HD bool fq9_is_zero_mod_p(const Fq9& a)
{
    static_assert(J <= 12, "bound too large");
    // prefilter: V = j*p has the low limb j*p mod 2^29, i.e. l[0] * p^-1 = j (mod 2^29) -- one multiplication instead of
    // J comparisons (round 5)
    if (((a.l[0] * Fq9C::PINV) & Fq9C::MASK) >= (uint32_t)J) return false;
    // slow path: subtract p until the value is below p (at most J-1 times), then test for zero
    Fq9 v = a;
    for (int j = 0; j < J; j++) {
        // v >= p ?
        bool ge = true;
        for (int i = 8; i >= 0; i--) {
            if (v.l[i] != Fq9C::P[i]) {
                ge = v.l[i] > Fq9C::P[i];
                break;
            }
        }
        if (!ge) break;
        int32_t c = 0;
        for (int i = 0; i < 9; i++) {
            int32_t t = (int32_t)v.l[i] - (int32_t)Fq9C::P[i] + c;
            if (i < 8) {
                v.l[i] = (uint32_t)t & Fq9C::MASK;
                c      = t >> 29;
            } else {
                v.l[8] = (uint32_t)t;
            }
        }
    }
    return fq9_limbs_zero(v);
}

// ---- packing: 9 x 29-bit limbs <-> 8 x 32-bit words (value must be < 2^256)
K16_HD Fq9 fq9_unpack(const uint32_t w[8])
{
    Fq9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, wd = bit >> 5, sh = bit & 31;
        uint64_t  v = w[wd];
        if (wd + 1 < 8) v |= (uint64_t)w[wd + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & (i < 8 ? Fq9C::MASK : 0xffffffffu);
    }
    return r;
}
K16_HD void fq9_pack(uint32_t w[8], const Fq9& a)
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // word k covers bits [32k, 32k+32)
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int lo = 29 * i - 32 * k; // position of limb i relative to word k
            if (lo > -29 && lo < 32) v |= lo >= 0 ? ((uint64_t)a.l[i] << lo) : ((uint64_t)a.l[i] >> (-lo));
        }
        w[k] = (uint32_t)v;
    }
}

// canonical Montgomery (R = 2^256, 8 x u32, < p)  ->  Fq9 (R' domain, < 2p)
K16_HD Fq9 fq9_from_fq(const Fq& x)
{
    Fq9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Fq9C::K_IN[i];
    return fmul9(fq9_unpack(x.v), k);
}
// Fq9 (any bound <= 12p)  ->  canonical Montgomery Fq (< p)
K16_HD Fq fq9_to_fq(const Fq9& a)
{
    Fq9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Fq9C::K_OUT[i];
    Fq9 v = fmul9(a, k); // = x*R, < 2p
    Fq  r;
    fq9_pack(r.v, v);
    cond_sub_p<FqParams>(r.v);
    return r;
}

// a^(p-2): the inverse in the same R' domain (Fermat; square-and-multiply over the 254 bits of p - 2).
// a < 2p, a != 0 mod p; result < 2p.  Used where inversions are rare (building fixed-base window tables).
K16_HD Fq9 finv9(const Fq9& a)
{
    const uint32_t e[8] = {FqParams::P[0] - 2u, FqParams::P[1], FqParams::P[2], FqParams::P[3],
                           FqParams::P[4],      FqParams::P[5], FqParams::P[6], FqParams::P[7]}; // p - 2 (no borrow)
    Fq9 r = fq9_one();
    for (int bit = 253; bit >= 0; bit--) {
        r = fsqr9(r);
        if ((e[bit >> 5] >> (bit & 31)) & 1u) r = fmul9(r, a);
    }
    return r;
}

// partial reduction (used by Fq2n / Fr9 below and by the rare branches of the G1 formulas)
// v < 32p, normalised  ->  v - q*p with q = floor(v.l[8] / 3171407)  in [0, p * (1 + 2^-17)):
// v / 2^232 < (q + 1) * 3171407 and p / 2^232 > 3171406.3, so the remainder is below p (1 + (q + 1) * 2.2e-7) -- 7.1e-6 at
// q = 31, against 2^-17 = 7.6e-6 (the NTT passes reduce tile values of up to 32 r with it); the reciprocal estimate is exact
// or one less for top limbs below 2^27 (error t * 0.2 / 2^44 << 1), which the correction step settles.
template <class C>
K16_HD Fq9 fred9_t(const Fq9& v)
{
    const uint32_t t = v.l[8];
    uint32_t       q = (uint32_t)(((uint64_t)t * 5547123ull) >> 44); // floor(t / 3171407) or one less
    q += ((q + 1) * 3171407u <= t) ? 1u : 0u;
    Fq9     r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int64_t x = (int64_t)v.l[i] - (int64_t)((uint64_t)q * C::P[i]) + c;
        if (i < 8) {
            r.l[i] = (uint32_t)x & C::MASK;
            c      = x >> 29;
        } else {
            r.l[8] = (uint32_t)x;
        }
    }
    return r;
}

K16_HD Fq9 fred9(const Fq9& v) { return fred9_t<Fq9C>(v); }

// ------------------------------------------------------------------------------------------------
// G1 in XYZZ coordinates over Fq9.  Same formulas and the same exceptional-case order as
// bn254_curve.h (curve.cpp:91-458 of the reference).  Invariant of every stored point:
//     X < 8p,  Y < 4p,  ZZ < 2p,  ZZZ < 2p          (affine table entries: x, y < 2p)
// Bounds in the comments are multiples of p; every product obeys A*B <= 128.
// ------------------------------------------------------------------------------------------------
struct Aff9 {
    Fq9 x, y;
    K16_HD bool is_zero() const { return fq9_limbs_zero(x) && fq9_limbs_zero(y); } // (0,0) converts to exact zero limbs
};
struct Xyzz9 {
    Fq9 x, y, zz, zzz;
    K16_HD bool         is_zero() const { return fq9_is_zero_mod_p<2>(zz); }
    static K16_HD Xyzz9 zero() { return Xyzz9{fq9_one(), fq9_one(), fq9_zero(), fq9_zero()}; }
    static K16_HD Xyzz9 from_aff(const Aff9& a)
    {
        if (a.is_zero()) return zero();
        return Xyzz9{a.x, a.y, fq9_one(), fq9_one()};
    }
};

// curve.cpp:411-458
K16_HD Xyzz9 pdbl_aff9(const Aff9& p)
{
    if (p.is_zero()) return Xyzz9::zero();
    Fq9 U  = fdbl9(p.y);                            // 4
    Fq9 V  = fsqr9(U);                              // 16 -> 2
    Fq9 W  = fmul9(U, V);                           // 8  -> 2
    Fq9 S  = fmul9(p.x, V);                         // 4  -> 2
    Fq9 M  = fsqr9(p.x);                            // 4  -> 2
    M      = fadd9(fdbl9(M), M);                    // 6
    Fq9 X3 = fsub9<4>(fsqr9(M), fdbl9(S));          // 36 -> 2 ; - 2S (<4) -> 6
    Fq9 Y3 = fmul9_sum2(M, fsub9<8>(S, X3), W, fsub9<2>(fq9_zero(), p.y)); // M*(S-X3) + W*(-y): 6*10 + 2*2 -> 2
    return Xyzz9{X3, Y3, V, W};
}
// curve.cpp:340-396
K16_HD Xyzz9 pdbl9(const Xyzz9& p)
{
    if (p.is_zero()) return p;
    Fq9 U  = fdbl9(p.y);                            // 8
    Fq9 V  = fsqr9(U);                              // 64 -> 2
    Fq9 W  = fmul9(U, V);                           // 16 -> 2
    Fq9 S  = fmul9(p.x, V);                         // 16 -> 2
    Fq9 M  = fsqr9(p.x);                            // 64 -> 2
    M      = fadd9(fdbl9(M), M);                    // 6
    Fq9 X3 = fsub9<4>(fsqr9(M), fdbl9(S));          // < 6
    Fq9 Y3 = fmul9_sum2(M, fsub9<8>(S, X3), W, fsub9<4>(fq9_zero(), p.y)); // 6*10 + 2*4 -> 2
    return Xyzz9{X3, Y3, fmul9(V, p.zz), fmul9(W, p.zzz)};
}
// curve.cpp:185-250
K16_HD Xyzz9 padd_mixed9(const Xyzz9& p1, const Aff9& p2)
{
    if (p1.is_zero()) return Xyzz9::from_aff(p2);
    if (p2.is_zero()) return p1;
    Fq9 U2 = fmul9(p2.x, p1.zz);                    // 2*2 -> 2
    Fq9 S2 = fmul9(p2.y, p1.zzz);                   // 2
    Fq9 P  = fsub9<8>(U2, p1.x);                    // X1 < 8 -> P < 10
    Fq9 R  = fsub9<4>(S2, p1.y);                    // Y1 < 4 -> R < 6
    if (fq9_is_zero_mod_p<10>(P) && fq9_is_zero_mod_p<6>(R)) return pdbl_aff9(p2);
    Fq9 PP  = fsqr9(P);                             // 100 -> 2
    Fq9 PPP = fmul9(P, PP);                         // 20 -> 2
    Fq9 Q   = fmul9(p1.x, PP);                      // 16 -> 2
    Fq9 X3  = fsub9<4>(fsub9<2>(fsqr9(R), PPP), fdbl9(Q)); // 36 -> 2 ; -PPP -> 4 ; -2Q -> 8
    // Y3 = (Q - X3)*R - Y1*PPP as ONE reduction of two products: (Q - X3 + 8p) < 10, R < 6, (4p - Y1) <= 4, PPP < 2 -> 68/169
    Fq9 Y3  = fmul9_sum2(fsub9<8>(Q, X3), R, fsub9<4>(fq9_zero(), p1.y), PPP); // < 2
    return Xyzz9{X3, Y3, fmul9(p1.zz, PP), fmul9(p1.zzz, PPP)};
}
// ------------------------------------------------------------------------------------------------
// The bucket accumulator of the hot loop (round 5: the mixed addition on an "instruction diet").
// Same formulas, same branch order and the same (X, Y, ZZ, ZZZ) values as padd_mixed9 -- what changes is how the
// subtractions are carried out:
//   * the accumulator keeps W = +-Y with a flag (neg: W = -Y).  With R = S2 - Y1 and Y3 = R (Q - X3) - Y1 PPP:
//       neg = 1:  T = S2 + W = R,    W' = T (Q - X3) + W PPP =  Y3  ->  neg' = 0
//       neg = 0:  T = -S2 + W = -R,  W' = T (Q - X3) + W PPP = -Y3  ->  neg' = 1       (X3 = T^2 - PPP - 2Q either way)
//     so no addition negates Y1 (fsub9<4>(0, Y1) before), R is a plain sum, and the sign moves to the gathered row's y,
//     whose negation for the signed digits is there anyway: y is negated iff sign ^ neg ^ 1.
//   * that negation is LAZY (3p - y limb by limb, 2^29 lent downwards: no carries) -- it feeds one multiplication;
//   * X3 = T^2 + 6p - PPP - 2Q in ONE carry pass (three before), < 8p;
//   * Q - X3 + 10p lazy (limbs < 3 * 2^29), the lazy operand of the two-product reduction (column bound 45 * 2^58);
//   * the doubling test is a one-multiplication prefilter on P alone (P + (-P) needs no branch: the formulas give ZZ3 = 0).
// Invariant: X < 8p, W < 4p, ZZ, ZZZ < 2p, all normalised.
// ------------------------------------------------------------------------------------------------
struct Acc9 {
    Fq9      x, w, zz, zzz;
    uint32_t neg; // 1: w = -Y
    K16_HD bool is_zero() const { return fq9_is_zero_mod_p<2>(zz); }
    static K16_HD Acc9 zero() { return Acc9{fq9_one(), fq9_one(), fq9_zero(), fq9_zero(), 0u}; }
    static K16_HD Acc9 from_xyzz(const Xyzz9& p) { return Acc9{p.x, p.y, p.zz, p.zzz, 0u}; }
    K16_HD Xyzz9 to_xyzz() const { return Xyzz9{x, neg ? fsub9<4>(fq9_zero(), w) : w, zz, zzz}; } // W < 4p -> Y <= 4p
};
// K*p with 2^29 lent to each of the limbs 0..7 by the limb above (the same integer): b normalised with b.l[8] < (K*p).l[8]
// gives K*p - b limb by limb without a negative limb
template <int K>
K16_HD constexpr uint32_t fq9_lent_kp(int i)
{
    const uint32_t kp = K == 3 ? Fq9C::KP3[i] : (K == 10 ? Fq9C::KP10[i] : Fq9C::KP4[i]);
    return i == 0 ? kp + (1u << 29) : (i < 8 ? kp + (1u << 29) - 1u : kp - 1u);
}
// rr + 6p - ppp - 2q, normalised: needs ppp + 2q <= 6p; result < RR + 6
K16_HD Fq9 fq9_x3(const Fq9& rr, const Fq9& ppp, const Fq9& q)
{
    Fq9     r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t t = (int32_t)(rr.l[i] + Fq9C::KP6[i]) - (int32_t)(ppp.l[i] + 2u * q.l[i]) + c;
        if (i < 8) {
            r.l[i] = (uint32_t)t & Fq9C::MASK;
            c      = t >> 29;
        } else {
            r.l[8] = (uint32_t)t;
        }
    }
    return r;
}
#ifndef K16_FQ9_HOOK
#define K16_FQ9_HOOK(what, v) // tests/cpp/fq9_check.cpp looks at the intermediate values through this
#endif
// acc += (sign ? -row : row); row as it lies in the table (x, y < 2p, normalised; (0,0) = the point at infinity)
K16_HD void acc9_madd(Acc9& a, const Aff9& row, uint32_t sign)
{
    if (row.is_zero()) return;
    const bool negy = ((sign ^ a.neg ^ 1u) & 1u) != 0;
    Fq9        ym; // +-y, lazy: limbs < 2^30, value <= 3p
#pragma unroll
    for (int i = 0; i < 9; i++) ym.l[i] = negy ? fq9_lent_kp<3>(i) - row.y.l[i] : row.y.l[i];
    K16_FQ9_HOOK("ym", ym);
    if (a.is_zero()) { // infinity + row
        a.x   = row.x;
        a.w   = fadd9(ym, fq9_zero()); // normalise
        a.zz  = fq9_one();
        a.zzz = fq9_one();
        a.neg ^= 1u;
        return;
    }
    Fq9 U2 = fmul9(row.x, a.zz);                     // 2*2 -> 2
    Fq9 S  = fmul9(ym, a.zzz);                       // 3*2 -> 2
    Fq9 P  = fsub9<8>(U2, a.x);                      // X1 < 8 -> P < 10
    Fq9 T  = fadd9(S, a.w);                          // W < 4 -> T < 6
    if (fq9_maybe_zero_mod_p<10>(P)) {
        if (fq9_is_zero_mod_p<10>(P) && fq9_is_zero_mod_p<6>(T)) { // acc == +-row with the same sign: doubling
            // the signed row's y from ym (row.y itself is dead by now: nine registers less across the products above):
            // ym = +-y_e with the sign negy ^ sign, <= 3p; fred9 brings it below 2p for the doubling's bounds
            const Fq9 yn = fadd9(ym, fq9_zero());
            Xyzz9     d  = pdbl_aff9(Aff9{row.x, fred9(negy == ((sign & 1u) != 0) ? yn : fsub9<4>(fq9_zero(), yn))});
            a            = Acc9::from_xyzz(d);
            return;
        }
    }
    Fq9 PP  = fsqr9(P);                              // 100 -> 2
    Fq9 RR  = fsqr9(T);                              // 36 -> 2
    Fq9 PPP = fmul9(P, PP);                          // 20 -> 2
    Fq9 Q   = fmul9(a.x, PP);                        // 16 -> 2
    Fq9 X3  = fq9_x3(RR, PPP, Q);                    // < 8
    Fq9 D;                                           // Q - X3 + 10p, lazy: limbs < 3 * 2^29, value < 12p
#pragma unroll
    for (int i = 0; i < 9; i++) D.l[i] = Q.l[i] + fq9_lent_kp<10>(i) - X3.l[i];
    K16_FQ9_HOOK("D", D);
    Fq9 W3 = fmul9_sum2(D, T, a.w, PPP);             // 12*6 + 4*2 = 80 -> 2
    a.zz   = fmul9(a.zz, PP);
    a.zzz  = fmul9(a.zzz, PPP);
    a.x    = X3;
    a.w    = W3;
    a.neg ^= 1u;
}

// affine + affine -> XYZZ: the mixed addition above with ZZ1 = ZZZ1 = 1 (U2 = x2, S2 = y2, ZZ3 = PP, ZZZ3 = PPP), i.e.
// the same values as padd_mixed9(from_aff(a), b) for 4 multiplications less.  First add of every bucket segment.
K16_HD Xyzz9 padd_aff_aff9(const Aff9& a, const Aff9& b)
{
    if (a.is_zero()) return Xyzz9::from_aff(b);
    if (b.is_zero()) return Xyzz9::from_aff(a);
    Fq9 P = fsub9<2>(b.x, a.x);                     // 4
    Fq9 R = fsub9<2>(b.y, a.y);                     // 4
    if (fq9_is_zero_mod_p<4>(P) && fq9_is_zero_mod_p<4>(R)) return pdbl_aff9(b);
    Fq9 PP  = fsqr9(P);                             // 16 -> 2
    Fq9 PPP = fmul9(P, PP);                         // 8 -> 2
    Fq9 Q   = fmul9(a.x, PP);                       // 4 -> 2
    Fq9 X3  = fsub9<4>(fsub9<2>(fsqr9(R), PPP), fdbl9(Q)); // 8
    Fq9 Y3  = fmul9_sum2(fsub9<8>(Q, X3), R, fsub9<2>(fq9_zero(), a.y), PPP); // 10*4 + 2*2 -> 2
    return Xyzz9{X3, Y3, PP, PPP};
}
// curve.cpp:91-166
K16_HD Xyzz9 padd9(const Xyzz9& p1, const Xyzz9& p2)
{
    if (p1.is_zero()) return p2;
    if (p2.is_zero()) return p1;
    Fq9 U1 = fmul9(p1.x, p2.zz);                    // 16 -> 2
    Fq9 U2 = fmul9(p2.x, p1.zz);
    Fq9 S1 = fmul9(p1.y, p2.zzz);                   // 8 -> 2
    Fq9 S2 = fmul9(p2.y, p1.zzz);
    Fq9 P  = fsub9<2>(U2, U1);                      // 4
    Fq9 R  = fsub9<2>(S2, S1);                      // 4
    if (fq9_is_zero_mod_p<4>(P) && fq9_is_zero_mod_p<4>(R)) return pdbl9(p1);
    Fq9 PP  = fsqr9(P);                             // 16 -> 2
    Fq9 PPP = fmul9(P, PP);
    Fq9 Q   = fmul9(U1, PP);
    Fq9 X3  = fsub9<4>(fsub9<2>(fsqr9(R), PPP), fdbl9(Q)); // 8
    Fq9 Y3  = fmul9_sum2(fsub9<8>(Q, X3), R, fsub9<2>(fq9_zero(), S1), PPP); // 10*4 + 2*2 -> 2
    return Xyzz9{X3, Y3, fmul9(fmul9(p1.zz, p2.zz), PP), fmul9(fmul9(p1.zzz, p2.zzz), PPP)};
}

// conversions of whole points
K16_HD Aff9 aff9_from_canonical(const Aff<Fq>& a)
{
    return Aff9{fq9_from_fq(a.x), fq9_from_fq(a.y)}; // (0,0) -> exact zero limbs
}
K16_HD Xyzz<Fq> xyzz9_to_canonical(const Xyzz9& p)
{
    if (p.is_zero()) return Xyzz<Fq>::zero();
    return Xyzz<Fq>{fq9_to_fq(p.x), fq9_to_fq(p.y), fq9_to_fq(p.zz), fq9_to_fq(p.zzz)};
}
K16_HD Xyzz9 xyzz9_from_canonical(const Xyzz<Fq>& p)
{
    if (p.is_zero()) return Xyzz9::zero();
    return Xyzz9{fq9_from_fq(p.x), fq9_from_fq(p.y), fq9_from_fq(p.zz), fq9_from_fq(p.zzz)};
}


// ------------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1) over Fq9, with the simple invariant "every component < 2p (+ 2^-17 p)".
// Each operation ends in fred9, a partial reduction by a quotient estimated from the top limb
// (value / 2^232 against p / 2^232 = 3171406.3): one 9-limb multiply-subtract, ~15 % of a multiply.
// With the invariant in place the generic XYZZ formulas of bn254_curve.h apply unchanged (G2).
// ------------------------------------------------------------------------------------------------

struct Fq2n {
    Fq9 a, b;
    static K16_HD Fq2n zero() { return Fq2n{fq9_zero(), fq9_zero()}; }
    static K16_HD Fq2n one() { return Fq2n{fq9_one(), fq9_zero()}; }
    K16_HD bool        is_zero() const { return fq9_is_zero_mod_p<3>(a) && fq9_is_zero_mod_p<3>(b); }
};
K16_HD Fq2n fadd(const Fq2n& x, const Fq2n& y) { return Fq2n{fred9(fadd9(x.a, y.a)), fred9(fadd9(x.b, y.b))}; }
K16_HD Fq2n fsub(const Fq2n& x, const Fq2n& y) { return Fq2n{fred9(fsub9<4>(x.a, y.a)), fred9(fsub9<4>(x.b, y.b))}; } // 4p offset: slack for the 2p+eps invariant
K16_HD Fq2n fdbl(const Fq2n& x) { return Fq2n{fred9(fdbl9(x.a)), fred9(fdbl9(x.b))}; }
K16_HD Fq2n fneg(const Fq2n& x) { return fsub(Fq2n::zero(), x); }
// f2field.cpp:122-142 computes (a + bu)(c + du) = (ac - bd) + (ad + bc)u with Karatsuba's three products.  Here each
// component is ONE Montgomery reduction of two products (fmul9_sum2): 4 x 81 product terms + 2 x 81 reduction terms, the
// same 486 multiply-adds as 3 full multiplications, but none of Karatsuba's five additions / subtractions and no partial
// reduction afterwards: bounds 2*2 + 4*2 = 12 and 2*2 + 2*2 = 8 of the 128 allowed, results < 2p.
K16_HD Fq2n fmul(const Fq2n& x, const Fq2n& y)
{
    const Fq9 nb = fsub9<4>(fq9_zero(), x.b); // 4p - b  (b < 2p + eps)
    return Fq2n{fmul9_sum2(x.a, y.a, nb, y.b), fmul9_sum2(x.a, y.b, x.b, y.a)};
}
// f2field.cpp:144-158 (complex squaring)
K16_HD Fq2n fsqr(const Fq2n& x)
{
    Fq9 ra = fmul9(fadd9(x.a, x.b), fsub9<4>(x.a, x.b)); // 4 * 6 = 24
    return Fq2n{ra, fmul9(fdbl9(x.a), x.b)};              // 2ab as (2a) * b: 4 * 2 = 8, no reduction afterwards
}

K16_HD Fq2n fq2n_from_canonical(const Fq2& x) { return Fq2n{fq9_from_fq(x.a), fq9_from_fq(x.b)}; }
K16_HD Fq2  fq2n_to_canonical(const Fq2n& x) { return Fq2{fq9_to_fq(x.a), fq9_to_fq(x.b)}; }


// ------------------------------------------------------------------------------------------------
// Fr on the same representation (the NTT / polynomial chain).  An Fr9 is a 9-limb value congruent to
// x * 2^261 mod r, kept < 2r (+ 2^-17 r) between operations by frred9; "packed" = the same value as
// 8 x 32-bit words (what the kernels keep in HBM for a, b, c and the twiddle table).
// ------------------------------------------------------------------------------------------------
typedef Fq9 Fr9;
K16_HD Fr9 frmul9(const Fr9& a, const Fr9& b) { return fmul9_t<Fr9C>(a, b); }
K16_HD Fr9 fradd9(const Fr9& a, const Fr9& b) { return fred9_t<Fr9C>(fadd9(a, b)); }      // < 4r -> < 2r
K16_HD Fr9 frsub9(const Fr9& a, const Fr9& b) { return fred9_t<Fr9C>(fsub9_t<Fr9C, 2>(a, b)); } // b < 2r
K16_HD Fr9 fr9_load(const uint32_t w[8]) { return fq9_unpack(w); }
K16_HD void fr9_store(uint32_t w[8], const Fr9& a) { fq9_pack(w, a); }                    // a < 2^256
// canonical Montgomery Fr (R = 2^256) -> Fr9, and back (exact, canonical result)
K16_HD Fr9 fr9_from_fr(const Fr& x)
{
    Fr9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Fr9C::K_IN[i];
    return fmul9_t<Fr9C>(fq9_unpack(x.v), k);
}
K16_HD Fr fr9_to_fr(const Fr9& a)
{
    Fr9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = Fr9C::K_OUT[i];
    Fr9 v = fmul9_t<Fr9C>(a, k);
    Fr  r;
    fq9_pack(r.v, v);
    cond_sub_p<FrParams>(r.v);
    return r;
}
// Fr9 (x * 2^261) -> the STANDARD-form integer x, canonical (what the H MSM consumes: groth16.cpp:273)
K16_HD Fr fr9_to_standard(const Fr9& a)
{
    Fr9 one = fq9_zero();
    one.l[0] = 1;
    Fr9 v = fmul9_t<Fr9C>(a, one); // x * 2^261 / 2^261 = x, < 2r
    Fr  r;
    fq9_pack(r.v, v);
    cond_sub_p<FrParams>(r.v);
    return r;
}

} // namespace k16
