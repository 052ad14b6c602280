// bn254_field.h -- BN254 Fq / Fr Montgomery arithmetic, 8 x 32-bit limbs held in registers.
//
// gfx950 has a 32-bit integer multiplier (v_mad_u64_u32 = 32x32+64), so the reference's
// "4 x 64-bit limb" element (fr_element.hpp:13) is carried as 8 x u32; the byte layout in
// memory is identical (little-endian), so zkey / wtns buffers are consumed as they are.
// Semantics follow the reference's raw field ops (fq_raw_generic.cpp:12-40, 69-81, 108-149,
// 193-233): every result is canonical in [0, p); R = 2^256.
//
// The same source is compiled for the device (kernels) and for the host (blinding step,
// affine conversion, JSON) so the product has ONE field implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define K16_HD __host__ __device__ __forceinline__

namespace k16 {

struct FqParams {
    // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 (fq_raw_generic.cpp:6)
    static constexpr uint32_t P[8]  = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                       0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t NP    = 0xe4866389u; // -q^-1 mod 2^32 (low word of fq_raw_generic.cpp:8)
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
};
struct FrParams {
    // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 (fr_raw_generic.cpp:5)
    static constexpr uint32_t P[8]  = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                       0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t NP    = 0xefffffffu; // fr_raw_generic.cpp:7 (low word)
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
};

template <class PR>
struct Fp {
    uint32_t v[8];

    static K16_HD Fp zero()
    {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = 0;
        return r;
    }
    static K16_HD Fp one()
    {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = PR::ONE[i];
        return r;
    }
    static K16_HD Fp r2()
    {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = PR::R2[i];
        return r;
    }
    K16_HD bool is_zero() const
    {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i];
        return o == 0;
    }
    K16_HD bool operator==(const Fp& b) const
    {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
        return o == 0;
    }
};

// r = a - p if a >= p (a < 2p)
template <class PR>
K16_HD void cond_sub_p(uint32_t t[8])
{
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t[i] - PR::P[i] - br;
        d[i]       = (uint32_t)x;
        br         = (x >> 63) & 1;
    }
    if (!br) {
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = d[i];
    }
}

// fq_raw_generic.cpp:12-20.  Inputs canonical; a + b < 2p < 2^255 so no carry out of 256 bits.
template <class PR>
K16_HD Fp<PR> fadd(const Fp<PR>& a, const Fp<PR>& b)
{
    Fp<PR>   r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    cond_sub_p<PR>(r.v);
    return r;
}
// fq_raw_generic.cpp:32-40
template <class PR>
K16_HD Fp<PR> fsub(const Fp<PR>& a, const Fp<PR>& b)
{
    Fp<PR>   r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - br;
        r.v[i]     = (uint32_t)x;
        br         = (x >> 63) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c    = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)r.v[i] + (PR::P[i] & mask);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
// fq_raw_generic.cpp:69-81 : neg(0) = 0
template <class PR>
K16_HD Fp<PR> fneg(const Fp<PR>& a)
{
    return fsub(Fp<PR>::zero(), a);
}
template <class PR>
K16_HD Fp<PR> fdbl(const Fp<PR>& a)
{
    return fadd(a, a);
}

// fq_raw_generic.cpp:108-149 : Montgomery product, CIOS over 32-bit words.  p < 2^254 so the
// two carry chains of a CIOS round can be fused (top word of p has two spare bits): per round
//   (A,t0) = t0 + a0*bi ; m = t0*np ; (C,_) = t0 + m*p0
//   (A,tj) = tj + aj*bi + A ; (C,t[j-1]) = tj + m*pj + C ; t7 = A + C
template <class PR>
K16_HD Fp<PR> fmul(const Fp<PR>& a, const Fp<PR>& b)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    // host build (blinding step, Horner tail, affine/JSON): the same CIOS on 4 x 64-bit words
    typedef unsigned __int128 u128;
    uint64_t A[4], B[4], P[4], t[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
        P[i] = (uint64_t)PR::P[2 * i] | ((uint64_t)PR::P[2 * i + 1] << 32);
    }
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: x' = x * (2 + p*x)  (x = -p^-1)
    uint64_t np = (uint64_t)PR::NP;
    np          = np * (2 + P[0] * np);
    for (int i = 0; i < 4; i++) {
        u128     X = (u128)A[0] * B[i] + t[0];
        uint64_t m = (uint64_t)X * np;
        u128     C = (u128)m * P[0] + (uint64_t)X;
        X >>= 64;
        C >>= 64;
        for (int j = 1; j < 4; j++) {
            X += (u128)A[j] * B[i] + t[j];
            C += (u128)m * P[j] + (uint64_t)X;
            t[j - 1] = (uint64_t)C;
            X >>= 64;
            C >>= 64;
        }
        t[3] = (uint64_t)(X + C);
    }
    uint32_t r32[8];
    for (int i = 0; i < 4; i++) {
        r32[2 * i]     = (uint32_t)t[i];
        r32[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    cond_sub_p<PR>(r32);
    Fp<PR> r;
    for (int i = 0; i < 8; i++) r.v[i] = r32[i];
    return r;
#else
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
        uint32_t m = (uint32_t)A * PR::NP;
        uint64_t C = (uint64_t)m * PR::P[0] + (uint32_t)A;
        A >>= 32;
        C >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            A += (uint64_t)a.v[j] * b.v[i] + t[j];
            C += (uint64_t)m * PR::P[j] + (uint32_t)A;
            t[j - 1] = (uint32_t)C;
            A >>= 32;
            C >>= 32;
        }
        t[7] = (uint32_t)(A + C);
    }
    cond_sub_p<PR>(t);
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
#endif
}
template <class PR>
K16_HD Fp<PR> fsqr(const Fp<PR>& a)
{
    return fmul(a, a);
}
// fq_raw_generic.cpp:193-233
template <class PR>
K16_HD Fp<PR> to_mont(const Fp<PR>& a)
{
    return fmul(a, Fp<PR>::r2());
}
template <class PR>
K16_HD Fp<PR> from_mont(const Fp<PR>& a)
{
    Fp<PR> one = Fp<PR>::zero();
    one.v[0]   = 1;
    return fmul(a, one);
}
// a^e, e as 8 x u32 little-endian (fq.cpp:259-278)
template <class PR>
K16_HD Fp<PR> fpow(const Fp<PR>& a, const uint32_t e[8])
{
    Fp<PR> acc   = Fp<PR>::one();
    bool   found = false;
    for (int i = 255; i >= 0; i--) {
        bool bit = (e[i >> 5] >> (i & 31)) & 1;
        if (found) acc = fsqr(acc);
        if (bit) {
            acc   = found ? fmul(acc, a) : a;
            found = true;
        }
    }
    return acc;
}
// Montgomery form of a^-1 (fq.cpp:238-250 gives the same canonical value); inv(0) = 0.
template <class PR>
K16_HD Fp<PR> finv(const Fp<PR>& a)
{
    uint32_t e[8];
    uint64_t br = 2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)PR::P[i] - br;
        e[i]       = (uint32_t)x;
        br         = (x >> 63) & 1;
    }
    if (a.is_zero()) return a;
    return fpow(a, e);
}

// The same value as finv (Montgomery form of a^-1; inv(0) = 0) by the binary extended Euclidean algorithm instead of
// Fermat's 254 squarings + ~127 multiplications: ~500 shift / subtract steps on eight limbs, i.e. ~20 multiplication times
// instead of ~380 for ONE element.  For latency-critical single inversions (the wave-cooperative verifier); the batched
// kernels keep finv, whose lanes stay in lockstep.  x = aR, plain inverse y = x^-1 mod p, result a^-1 R = y * R^3 / R.
template <class PR>
__host__ __device__ __attribute__((noinline)) Fp<PR> finv_bgcd(const Fp<PR>& a)
{
    if (a.is_zero()) return a;
    uint32_t u[8], v[8], x1[8], x2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u[i]  = a.v[i];
        v[i]  = PR::P[i];
        x1[i] = i == 0 ? 1u : 0u;
        x2[i] = 0u;
    }
    auto is_one = [](const uint32_t* w) {
        uint32_t o = w[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 8; i++) o |= w[i];
        return o == 0;
    };
    auto shr1 = [](uint32_t* w, uint32_t top) { // (top:w) >> 1
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
        w[7] = (w[7] >> 1) | (top << 31);
    };
    auto half_mod = [&](uint32_t* x) { // x / 2 mod p, x < p
        uint32_t carry = 0;
        if (x[0] & 1u) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint64_t t = (uint64_t)x[i] + PR::P[i] + carry;
                x[i]       = (uint32_t)t;
                carry      = (uint32_t)(t >> 32);
            }
        }
        shr1(x, carry);
    };
    auto geq = [](const uint32_t* p, const uint32_t* q) {
        for (int i = 7; i >= 0; i--)
            if (p[i] != q[i]) return p[i] > q[i];
        return true;
    };
    auto sub = [](uint32_t* p, const uint32_t* q) -> uint32_t { // p -= q, returns the borrow
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)p[i] - q[i] - br;
            p[i]       = (uint32_t)t;
            br         = (uint32_t)(t >> 63);
        }
        return br;
    };
    auto sub_mod = [&](uint32_t* p, const uint32_t* q) { // p = p - q mod P
        if (sub(p, q)) {
            uint32_t carry = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint64_t t = (uint64_t)p[i] + PR::P[i] + carry;
                p[i]       = (uint32_t)t;
                carry      = (uint32_t)(t >> 32);
            }
        }
    };
#pragma clang loop unroll(disable)
    for (int guard = 0; guard < 1100 && !is_one(u) && !is_one(v); guard++) {
        if (!(u[0] & 1u)) {
            shr1(u, 0);
            half_mod(x1);
        } else if (!(v[0] & 1u)) {
            shr1(v, 0);
            half_mod(x2);
        } else if (geq(u, v)) {
            (void)sub(u, v);
            sub_mod(x1, x2);
        } else {
            (void)sub(v, u);
            sub_mod(x2, x1);
        }
    }
    Fp<PR> y;
    const uint32_t* src = is_one(u) ? x1 : x2;
#pragma unroll
    for (int i = 0; i < 8; i++) y.v[i] = src[i];
    const Fp<PR> r2 = Fp<PR>::r2();
    return fmul(y, fmul(r2, r2)); // y * R^3 / R
}

typedef Fp<FqParams> Fq;
typedef Fp<FrParams> Fr;

// ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)  (f2field.cpp)
struct Fq2 {
    Fq a, b;
    static K16_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static K16_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    K16_HD bool       is_zero() const { return a.is_zero() && b.is_zero(); }
    K16_HD bool       operator==(const Fq2& o) const { return a == o.a && b == o.b; }
};
K16_HD Fq2 fadd(const Fq2& x, const Fq2& y) { return Fq2{fadd(x.a, y.a), fadd(x.b, y.b)}; }
K16_HD Fq2 fsub(const Fq2& x, const Fq2& y) { return Fq2{fsub(x.a, y.a), fsub(x.b, y.b)}; }
K16_HD Fq2 fneg(const Fq2& x) { return Fq2{fneg(x.a), fneg(x.b)}; }
K16_HD Fq2 fdbl(const Fq2& x) { return Fq2{fdbl(x.a), fdbl(x.b)}; }
// f2field.cpp:122-142 (Karatsuba, non-residue -1)
K16_HD Fq2 fmul(const Fq2& x, const Fq2& y)
{
    Fq aa = fmul(x.a, y.a);
    Fq bb = fmul(x.b, y.b);
    Fq s  = fmul(fadd(x.a, x.b), fadd(y.a, y.b));
    return Fq2{fsub(aa, bb), fsub(fsub(s, aa), bb)};
}
// f2field.cpp:144-158 (complex squaring)
K16_HD Fq2 fsqr(const Fq2& x)
{
    Fq ab = fmul(x.a, x.b);
    Fq ra = fmul(fadd(x.a, x.b), fsub(x.a, x.b));
    return Fq2{ra, fdbl(ab)};
}
// f2field.cpp:178-190
K16_HD Fq2 finv(const Fq2& x)
{
    Fq t = finv(fadd(fsqr(x.a), fsqr(x.b)));
    return Fq2{fmul(x.a, t), fneg(fmul(x.b, t))};
}

} // namespace k16
