// bn254_curve.h -- y^2 = x^3 + b (a = 0) in XYZZ coordinates over F = Fq (G1) or Fq2 (G2).
//
// Same formulas and the same exceptional-case order as the reference's Curve<BaseField>
// (rust-rapidsnark/rapidsnark/src/curve.cpp): add :91-166, mixed add :185-250, dbl :340-396,
// affine dbl :411-458, infinity conventions :39-44 / :532-539, to-affine :565-576.
// Affine infinity is (0,0); XYZZ infinity is zz == 0 (canonical (1,1,0,0)).
#pragma once
#include "bn254_field.h"

namespace k16 {

template <class F>
struct Aff {
    F x, y;
    K16_HD bool is_zero() const { return x.is_zero() && y.is_zero(); }
};
template <class F>
struct Xyzz {
    F x, y, zz, zzz;
    K16_HD bool        is_zero() const { return zz.is_zero(); }
    static K16_HD Xyzz zero() { return Xyzz{F::one(), F::one(), F::zero(), F::zero()}; }
    static K16_HD Xyzz from_aff(const Aff<F>& a)
    {
        if (a.is_zero()) return zero();
        return Xyzz{a.x, a.y, F::one(), F::one()};
    }
};

// curve.cpp:411-458
template <class F>
K16_HD Xyzz<F> pdbl_aff(const Aff<F>& p)
{
    if (p.is_zero()) return Xyzz<F>::zero();
    F U  = fdbl(p.y);
    F V  = fsqr(U);
    F W  = fmul(U, V);
    F S  = fmul(p.x, V);
    F M  = fsqr(p.x);
    M    = fadd(fdbl(M), M);
    F X3 = fsub(fsub(fsqr(M), S), S);
    F Y3 = fsub(fmul(M, fsub(S, X3)), fmul(W, p.y));
    return Xyzz<F>{X3, Y3, V, W};
}
// curve.cpp:340-396 (a = 0)
template <class F>
K16_HD Xyzz<F> pdbl(const Xyzz<F>& p)
{
    if (p.is_zero()) return p;
    F U  = fdbl(p.y);
    F V  = fsqr(U);
    F W  = fmul(U, V);
    F S  = fmul(p.x, V);
    F M  = fsqr(p.x);
    M    = fadd(fdbl(M), M);
    F X3 = fsub(fsub(fsqr(M), S), S);
    F Y3 = fsub(fmul(M, fsub(S, X3)), fmul(W, p.y));
    return Xyzz<F>{X3, Y3, fmul(V, p.zz), fmul(W, p.zzz)};
}
// curve.cpp:185-250 (EFD madd-2008-s)
template <class F>
K16_HD Xyzz<F> padd_mixed(const Xyzz<F>& p1, const Aff<F>& p2)
{
    if (p1.is_zero()) return Xyzz<F>::from_aff(p2);
    if (p2.is_zero()) return p1;
    F U2 = fmul(p2.x, p1.zz);
    F S2 = fmul(p2.y, p1.zzz);
    F P  = fsub(U2, p1.x);
    F R  = fsub(S2, p1.y);
    if (P.is_zero() && R.is_zero()) return pdbl_aff(p2);
    F PP  = fsqr(P);
    F PPP = fmul(P, PP);
    F Q   = fmul(p1.x, PP);
    F X3  = fsub(fsub(fsub(fsqr(R), PPP), Q), Q);
    F Y3  = fsub(fmul(fsub(Q, X3), R), fmul(p1.y, PPP));
    return Xyzz<F>{X3, Y3, fmul(p1.zz, PP), fmul(p1.zzz, PPP)};
}
// curve.cpp:91-166 (EFD add-2008-s)
template <class F>
K16_HD Xyzz<F> padd(const Xyzz<F>& p1, const Xyzz<F>& p2)
{
    if (p1.is_zero()) return p2;
    if (p2.is_zero()) return p1;
    F U1 = fmul(p1.x, p2.zz);
    F U2 = fmul(p2.x, p1.zz);
    F S1 = fmul(p1.y, p2.zzz);
    F S2 = fmul(p2.y, p1.zzz);
    F P  = fsub(U2, U1);
    F R  = fsub(S2, S1);
    if (P.is_zero() && R.is_zero()) return pdbl(p1);
    F PP  = fsqr(P);
    F PPP = fmul(P, PP);
    F Q   = fmul(U1, PP);
    F X3  = fsub(fsub(fsub(fsqr(R), PPP), Q), Q);
    F Y3  = fsub(fmul(fsub(Q, X3), R), fmul(S1, PPP));
    return Xyzz<F>{X3, Y3, fmul(fmul(p1.zz, p2.zz), PP), fmul(fmul(p1.zzz, p2.zzz), PPP)};
}
template <class F>
K16_HD Xyzz<F> pneg(const Xyzz<F>& p)
{
    return Xyzz<F>{p.x, fneg(p.y), p.zz, p.zzz};
}
template <class F>
K16_HD Aff<F> pneg(const Aff<F>& p)
{
    return Aff<F>{p.x, fneg(p.y)};
}
// curve.cpp:565-576
// (host: ONE inversion, by the binary extended Euclid -- the three conversions at the end of a proof were six Fermat
// inversions, ~0.1 ms on the critical path; the inverse is unique, so the result is the same canonical value)
K16_HD Fq finv_once(const Fq& x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return finv(x);
#else
    return finv_bgcd(x);
#endif
}
K16_HD Fq2 finv_once(const Fq2& x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return finv(x);
#else
    const Fq t = finv_bgcd(fadd(fsqr(x.a), fsqr(x.b))); // f2field.cpp:178-190: (a - bu) / (a^2 + b^2)
    return Fq2{fmul(x.a, t), fneg(fmul(x.b, t))};
#endif
}
template <class F>
K16_HD Aff<F> to_affine(const Xyzz<F>& p)
{
    if (p.is_zero()) return Aff<F>{F::zero(), F::zero()};
#ifdef __HIP_DEVICE_COMPILE__
    return Aff<F>{fmul(p.x, finv(p.zz)), fmul(p.y, finv(p.zzz))};
#else
    const F t = finv_once(fmul(p.zz, p.zzz)); // 1 / zz = t zzz, 1 / zzz = t zz
    return Aff<F>{fmul(p.x, fmul(t, p.zzz)), fmul(p.y, fmul(t, p.zz))};
#endif
}

// Single scalar multiplication (Curve::mulByScalar, curve.hpp:195-207 -> exp.hpp:9-31): NAF walk,
// MSB first, dbl / add / sub.  scalar: 32 bytes little-endian, any 256-bit value.
template <class F>
K16_HD Xyzz<F> pmul_scalar(const Xyzz<F>& base, const uint8_t scalar[32])
{
    // NAF digits of a 256-bit k need up to 257 positions
    int8_t   naf[260];
    uint32_t k[9];
    for (int i = 0; i < 8; i++)
        k[i] = (uint32_t)scalar[4 * i] | ((uint32_t)scalar[4 * i + 1] << 8) | ((uint32_t)scalar[4 * i + 2] << 16) |
               ((uint32_t)scalar[4 * i + 3] << 24);
    k[8] = 0;
    for (int i = 0; i < 258; i++) {
        int8_t d = 0;
        if (k[0] & 1) {
            if (k[0] & 2) {
                d = -1;
                for (int j = 0; j < 9; j++) {
                    if (++k[j] != 0) break;
                }
            } else {
                d = 1;
                k[0] &= ~1u;
            }
        }
        naf[i] = d;
        for (int j = 0; j < 8; j++) k[j] = (k[j] >> 1) | (k[j + 1] << 31);
        k[8] >>= 1;
    }
    Xyzz<F> nb  = pneg(base);
    Xyzz<F> acc = Xyzz<F>::zero();
    int     i   = 257;
    while (i >= 0 && naf[i] == 0) i--;
    for (; i >= 0; i--) {
        acc = pdbl(acc);
        if (naf[i] == 1)
            acc = padd(acc, base);
        else if (naf[i] == -1)
            acc = padd(acc, nb);
    }
    return acc;
}

typedef Aff<Fq>   G1Aff;
typedef Xyzz<Fq>  G1Xyzz;
typedef Aff<Fq2>  G2Aff;
typedef Xyzz<Fq2> G2Xyzz;

} // namespace k16
