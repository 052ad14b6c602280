// tests/cpp/fq9_check.cpp -- host-side cross-check of the radix-2^29 field / G1 code (bn254_fq9.h)
// against the canonical 8x32 implementation (bn254_field.h / bn254_curve.h), including the value
// bounds the lazy reduction relies on.  Built and run by tests/test_fq9_host.py (no GPU needed: both
// headers are __host__ __device__).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "bn254_curve.h"
// intermediate values of acc9_madd (lazy limbs): the header calls this hook when it is defined
namespace k16 { struct Fq9; }
static void fq9_hook(const char* what, const k16::Fq9& v);
#define K16_FQ9_HOOK(what, v) fq9_hook(what, v)
#include "bn254_fq9.h"
using namespace k16;

static uint64_t sm = 0x1234567;
static uint64_t rnd()
{
    sm += 0x9E3779B97F4A7C15ull;
    uint64_t z = sm;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static Fq rand_fq()
{
    Fq x;
    for (int i = 0; i < 8; i += 2) {
        uint64_t r = rnd();
        x.v[i] = (uint32_t)r;
        x.v[i + 1] = (uint32_t)(r >> 32);
    }
    x.v[7] &= 0x0fffffff; // < 2^252 < p
    return x;
}
// v < K*p ?  (v normalised)
static bool below_kp(const Fq9& v, int K)
{
    uint64_t kp[9], c = 0;
    for (int i = 0; i < 9; i++) {
        uint64_t t = (uint64_t)Fq9C::P[i] * K + c;
        if (i < 8) { kp[i] = t & Fq9C::MASK; c = t >> 29; } else kp[i] = t;
    }
    for (int i = 8; i >= 0; i--) {
        if (v.l[i] != kp[i]) return v.l[i] < kp[i];
    }
    return false;
}
static bool normalised(const Fq9& v)
{
    for (int i = 0; i < 8; i++) if (v.l[i] >> 29) return false;
    return true;
}
static int fails = 0;
static int hook_calls = 0;
static void fq9_hook(const char* what, const k16::Fq9& v)
{
    // "ym": +-y lazy, limbs < 2^30 (no wrapped negative limb); "D": Q - X3 + 10p lazy, limbs < 3 * 2^29
    const uint32_t lim = what[0] == 'y' ? (1u << 30) : 3u * (1u << 29);
    hook_calls++;
    for (int i = 0; i < 9; i++)
        if (v.l[i] >= lim) { fails++; if (fails < 20) printf("FAIL lazy limb bound of %s: limb %d = %08x\n", what, i, v.l[i]); }
}
#define CHECK(c, msg) do { if (!(c)) { fails++; if (fails < 20) printf("FAIL %s (line %d)\n", msg, __LINE__); } } while (0)

static bool pt_bounds_ok(const Xyzz9& p)
{
    return normalised(p.x) && normalised(p.y) && normalised(p.zz) && normalised(p.zzz) && below_kp(p.x, 8) &&
           below_kp(p.y, 4) && below_kp(p.zz, 2) && below_kp(p.zzz, 2);
}
static bool same_point_repr(const Xyzz9& a, const G1Xyzz& b)
{
    G1Xyzz c = xyzz9_to_canonical(a);
    if (b.is_zero()) return a.is_zero();
    return c.x == b.x && c.y == b.y && c.zz == b.zz && c.zzz == b.zzz;
}

int main()
{
    // field ops
    for (int it = 0; it < 20000; it++) {
        Fq a = rand_fq(), b = rand_fq();
        if (it == 0) a = Fq::zero();
        if (it == 1) { a = Fq::zero(); b = Fq::zero(); }
        if (it == 2) a = Fq::one();
        if (it == 3) { for (int i = 0; i < 8; i++) a.v[i] = FqParams::P[i]; a.v[0] -= 1; b = a; } // p-1
        Fq9 A = fq9_from_fq(a), B = fq9_from_fq(b);
        CHECK(normalised(A) && below_kp(A, 2), "from_fq bound");
        CHECK(fq9_to_fq(A) == a, "roundtrip");
        CHECK(fq9_to_fq(fmul9(A, B)) == fmul(a, b), "mul");
        CHECK(fq9_to_fq(fadd9(A, B)) == fadd(a, b), "add");
        CHECK(fq9_to_fq(fsub9<2>(A, B)) == fsub(a, b), "sub2");
        CHECK(fq9_to_fq(fsub9<8>(A, B)) == fsub(a, b), "sub8");
        Fq9 big = fadd9(fadd9(fadd9(A, B), fadd9(A, B)), fadd9(A, A)); // < 12p
        CHECK(fq9_to_fq(fmul9(big, B)) == fmul(fadd(fadd(fadd(a, b), fadd(a, b)), fadd(a, a)), b), "mul with lazy operand");
        CHECK(fq9_to_fq(fsqr9(A)) == fmul(a, a) && below_kp(fsqr9(A), 2) && normalised(fsqr9(A)), "sqr");
        CHECK(fq9_to_fq(fsqr9(big)) == fmul(fadd(fadd(fadd(a, b), fadd(a, b)), fadd(a, a)), fadd(fadd(fadd(a, b), fadd(a, b)), fadd(a, a))) && below_kp(fsqr9(big), 2), "sqr of a lazy operand (< 12p)");
        {
            Fq9 s2 = fmul9_sum2(big, fadd9(A, A), fsub9<4>(fq9_zero(), B), A); // 12*2... bounds 12*4 + 4*2 = 56
            Fq  w  = fsub(fmul(fadd(fadd(fadd(a, b), fadd(a, b)), fadd(a, a)), fadd(a, a)), fmul(b, a));
            CHECK(fq9_to_fq(s2) == w && below_kp(s2, 2) && normalised(s2), "a*b + c*d with one reduction");
        }
        if (it < 300 && !a.is_zero()) {
            Fq9 inv = finv9(A);
            CHECK(below_kp(inv, 2) && normalised(inv), "finv9 bound");
            CHECK(fq9_to_fq(fmul9(inv, A)) == Fq::one(), "finv9: a * a^-1 == 1");
        }
        CHECK(fq9_is_zero_mod_p<10>(fsub9<8>(A, A)), "x - x == 0 mod p");
        CHECK(fq9_is_zero_mod_p<4>(A) == a.is_zero(), "is_zero");
    }
    // lazy-limb forms of the NTT butterflies (ntt.hip): fadd9_lazy / fsub9_lazy4_t feeding a multiplication or a normalising
    // addition / subtraction, with the left operand up to 30p and worst-case limb patterns (all limbs 2^29 - 1 / 0)
    for (int it = 0; it < 20000; it++) {
        Fq a = rand_fq(), b = rand_fq(), w = rand_fq();
        Fq9 A = fq9_from_fq(a), B = fq9_from_fq(b), W = fq9_from_fq(w);
        Fq  big = a;
        Fq9 BIG = A;
        const int extra = it % 15; // BIG = A + 2 * extra * p  (same field element)
        {
            Fq9 pp;
            for (int i = 0; i < 9; i++) pp.l[i] = Fq9C::P[i];
            for (int j = 0; j < 2 * extra; j++) BIG = fadd9(BIG, pp);
        }
        if (it % 7 == 3) { // B with every low limb at its maximum (still < 2p: top limb 0x30644e < top limb of 2p)
            for (int i = 0; i < 8; i++) B.l[i] = Fq9C::MASK;
            B.l[8] = 0x30644e;
            b      = fq9_to_fq(B);
        }
        if (it % 7 == 5) { // BIG with all-zero low limbs (the difference then lives on the lent 2^29s alone)
            for (int i = 0; i < 8; i++) BIG.l[i] = 0;
            big = fq9_to_fq(BIG);
        }
        Fq9 s = fadd9_lazy(BIG, B), d = fsub9_lazy4_t<Fq9C>(BIG, B);
        bool limbs_ok = true;
        for (int i = 0; i < 9; i++) limbs_ok = limbs_ok && s.l[i] < (1u << 30) + (1u << 27) && d.l[i] < 3u * (1u << 29);
        CHECK(limbs_ok, "lazy limb bounds");
        CHECK(fq9_to_fq(fmul9(W, s)) == fmul(w, fadd(big, b)), "w * (a + b), lazy sum");
        CHECK(fq9_to_fq(fmul9(W, d)) == fmul(w, fsub(big, b)), "w * (a - b), lazy difference");
        Fq9 q = fmul9(W, A); // a fresh product (< 2p), the other operand of the second butterfly stage
        CHECK(normalised(fadd9(s, q)) && fq9_to_fq(fmul9(W, fadd9(s, q))) == fmul(w, fadd(fadd(big, b), fmul(w, a))), "lazy sum + product");
        CHECK(normalised(fadd9(d, q)) && fq9_to_fq(fmul9(W, fadd9(d, q))) == fmul(w, fadd(fsub(big, b), fmul(w, a))), "lazy difference + product");
        CHECK(normalised(fsub9<2>(s, q)) && fq9_to_fq(fmul9(W, fsub9<2>(s, q))) == fmul(w, fsub(fadd(big, b), fmul(w, a))), "lazy sum - product");
        CHECK(normalised(fsub9<2>(d, q)) && fq9_to_fq(fmul9(W, fsub9<2>(d, q))) == fmul(w, fsub(fsub(big, b), fmul(w, a))), "lazy difference - product");
    }
    // curve ops: random walks mixing madd / add / dbl with the exceptional cases
    G1Aff g;
    g.x = Fq::one();
    g.y = fdbl(Fq::one());
    const int NP = 64;
    G1Aff tab[NP];
    Aff9  tab9[NP];
    {
        G1Xyzz acc = G1Xyzz::zero();
        for (int i = 0; i < NP; i++) {
            acc     = padd_mixed(acc, g);
            tab[i]  = to_affine(acc);
            tab9[i] = aff9_from_canonical(tab[i]);
        }
        tab[7] = G1Aff{Fq::zero(), Fq::zero()}; // a (0,0) base
        tab9[7] = aff9_from_canonical(tab[7]);
        CHECK(tab9[7].is_zero(), "(0,0) converts to zero limbs");
    }
    for (int walk = 0; walk < 200; walk++) {
        G1Xyzz c = G1Xyzz::zero(), c2 = G1Xyzz::zero();
        Xyzz9  n = Xyzz9::zero(), n2 = Xyzz9::zero();
        for (int step = 0; step < 60; step++) {
            int op = (int)(rnd() % 8), k = (int)(rnd() % NP);
            if (step == 1) { op = 0; k = 3; }
            if (step == 2) { op = 0; k = 3; }               // maybe P + P
            if (op <= 3) { c = padd_mixed(c, tab[k]); n = padd_mixed9(n, tab9[k]); }
            else if (op == 4) { c = pdbl(c); n = pdbl9(n); }
            else if (op == 5) { c2 = padd(c2, c); n2 = padd9(n2, n); }
            else if (op == 6) { c = padd(c, c2); n = padd9(n, n2); }
            else { // P + (-P) and P + P through the projective add
                G1Xyzz neg = pneg(c);
                Xyzz9  neg9 = xyzz9_from_canonical(neg);
                G1Xyzz z = padd(c, neg);
                Xyzz9  z9 = padd9(n, neg9);
                CHECK(z.is_zero() && z9.is_zero(), "P + (-P)");
                c = padd(c, c); n = padd9(n, n);
            }
            CHECK(pt_bounds_ok(n) && pt_bounds_ok(n2), "point bounds");
            CHECK(same_point_repr(n, c), "XYZZ representation equal after canonicalisation");
            CHECK(same_point_repr(n2, c2), "XYZZ (second accumulator)");
        }
    }
    // affine + affine (first add of a bucket segment): same XYZZ values as the mixed add from from_aff
    for (int i = 0; i < NP; i++)
        for (int k = 0; k < NP; k++) {
            G1Xyzz w = padd_mixed(G1Xyzz::from_aff(tab[i]), tab[k]);
            Xyzz9  n = padd_aff_aff9(tab9[i], tab9[k]);
            CHECK(pt_bounds_ok(n), "aff+aff bounds");
            CHECK(same_point_repr(n, w), "aff+aff == madd(from_aff)");
            G1Aff ng = pneg(tab[k]);
            if (i == k && k != 7) CHECK(padd_aff_aff9(tab9[i], aff9_from_canonical(ng)).is_zero(), "aff + (-aff)");
        }
    // acc == table point -> doubling branch of the mixed add; acc == -table point -> infinity
    for (int k = 0; k < NP; k++) {
        if (k == 7) continue;
        G1Xyzz c = G1Xyzz::from_aff(tab[k]);
        Xyzz9  n = Xyzz9::from_aff(tab9[k]);
        CHECK(same_point_repr(padd_mixed9(n, tab9[k]), padd_mixed(c, tab[k])), "madd P+P");
        G1Aff  ng = pneg(tab[k]);
        Aff9   ng9 = aff9_from_canonical(ng);
        CHECK(padd_mixed9(n, ng9).is_zero() && padd_mixed(c, ng).is_zero(), "madd P+(-P)");
        // same with a non-trivial zz
        G1Xyzz c3 = padd_mixed(padd_mixed(c, tab[(k + 1) % NP == 7 ? 8 : (k + 1) % NP]), pneg(tab[(k + 1) % NP == 7 ? 8 : (k + 1) % NP]));
        Xyzz9  n3 = xyzz9_from_canonical(c3);
        CHECK(same_point_repr(padd_mixed9(n3, tab9[k]), padd_mixed(c3, tab[k])), "madd P+P (zz != 1)");
    }
    // ---- the bucket accumulator (acc9_madd: W = +-Y with a flag, lazy subtractions): the same XYZZ representation as the
    // reference's mixed addition for every step of random walks over signed rows, incl. (0,0) rows, P + P, P + (-P),
    // infinity + P, and rows / accumulators at the edges of their bounds
    {
        auto acc_ok = [&](const Acc9& a) {
            return normalised(a.x) && normalised(a.w) && normalised(a.zz) && normalised(a.zzz) && below_kp(a.x, 8) &&
                   below_kp(a.w, 4) && below_kp(a.zz, 2) && below_kp(a.zzz, 2) && a.neg <= 1u;
        };
        // table rows with y in the top sliver [top limb of 2p, 2p): y + p has the same residue and is < 2p for y < p
        Aff9 tabhi[NP];
        for (int k = 0; k < NP; k++) {
            tabhi[k] = tab9[k];
            if (k == 7) continue;
            Fq9 pp;
            for (int i = 0; i < 9; i++) pp.l[i] = Fq9C::P[i];
            Fq9 yc = fq9_from_fq(tab[k].y); // < 2p
            Fq9 y1 = fadd9(yc, pp);
            if (below_kp(yc, 1) && below_kp(y1, 2)) tabhi[k].y = y1;
        }
        for (int walk = 0; walk < 400; walk++) {
            G1Xyzz c = G1Xyzz::zero();
            Acc9   a = Acc9::zero();
            a.neg    = walk & 1u;
            for (int step = 0; step < 80; step++) {
                int      k    = (int)(rnd() % NP);
                uint32_t sign = (uint32_t)(rnd() & 1u);
                const Aff9* t9 = (walk % 3 == 2) ? tabhi : tab9;
                int mode = (int)(rnd() % 16);
                if (step == 1 && walk % 4 == 0) mode = 12; // P + P right after infinity + P
                if (mode == 12 && !c.is_zero()) {
                    // add the accumulator's own affine value: doubling through the mixed addition (zz != 1 in general)
                    G1Aff self = to_affine(c);
                    Aff9  s9   = aff9_from_canonical(self);
                    G1Aff arg  = sign ? pneg(self) : self; // row such that (sign ? -row : row) == self
                    Aff9  a9   = sign ? aff9_from_canonical(arg) : s9;
                    c = padd_mixed(c, self);
                    acc9_madd(a, a9, sign);
                } else if (mode == 13 && !c.is_zero()) {
                    // add minus the accumulator: infinity through the general formulas
                    G1Aff self = to_affine(c);
                    G1Aff ng   = pneg(self);
                    Aff9  a9   = sign ? aff9_from_canonical(self) : aff9_from_canonical(ng);
                    c = padd_mixed(c, ng);
                    acc9_madd(a, a9, sign);
                    CHECK(a.is_zero() && c.is_zero(), "acc9: P + (-P)");
                } else {
                    G1Aff arg = sign ? pneg(tab[k]) : tab[k];
                    c = padd_mixed(c, arg);
                    acc9_madd(a, t9[k], sign);
                }
                CHECK(acc_ok(a), "acc9 bounds");
                Xyzz9 x = a.to_xyzz();
                CHECK(normalised(x.y) && (below_kp(x.y, 4) || fq9_to_fq(x.y) == Fq::zero()), "acc9 -> xyzz Y bound");
                CHECK(same_point_repr(x, c), "acc9: XYZZ representation equal to the reference's mixed addition");
            }
        }
        // from an XYZZ point at its bounds (what the aff + aff first addition hands over): X up to 8p, Y up to 4p
        for (int it = 0; it < 2000; it++) {
            int    k = (int)(rnd() % NP), k2 = (int)(rnd() % NP);
            if (k == 7 || k2 == 7) continue;
            G1Xyzz c = padd_mixed(padd_mixed(G1Xyzz::zero(), tab[k]), tab[(k + 5) % NP == 7 ? 9 : (k + 5) % NP]);
            Xyzz9  n = xyzz9_from_canonical(c);
            Fq9 pp;
            for (int i = 0; i < 9; i++) pp.l[i] = Fq9C::P[i];
            for (int j = 0; j < (it % 6); j++) n.x = fadd9(n.x, pp);  // < 2p + 5p
            for (int j = 0; j < (it % 2); j++) n.y = fadd9(n.y, pp);  // < 3p
            Acc9 a = Acc9::from_xyzz(n);
            uint32_t sign = it & 1u;
            acc9_madd(a, tab9[k2], sign);
            G1Xyzz w = padd_mixed(c, sign ? pneg(tab[k2]) : tab[k2]);
            CHECK(same_point_repr(a.to_xyzz(), w), "acc9 from an XYZZ point at its bounds");
        }
        // the lazy forms on their own, at the edges: y with the top limb of 2p (3p - y must not lend from a negative top
        // limb), all-ones / all-zero low limbs; X3 just below 8p
        for (int it = 0; it < 4000; it++) {
            Fq  zc = rand_fq();
            Fq9 Z  = fq9_from_fq(zc), y;
            for (int i = 0; i < 8; i++) y.l[i] = (it & 1) ? Fq9C::MASK : ((it & 2) ? 0u : (uint32_t)rnd() & Fq9C::MASK);
            y.l[8] = (it % 5 == 0) ? Fq9C::KP2[8] : (uint32_t)(rnd() % (Fq9C::KP2[8] + 1u));
            if (!below_kp(y, 2)) { for (int i = 0; i < 8; i++) y.l[i] = 0; } // top limb of 2p with low limbs below 2p's
            Fq9 ym;
            for (int i = 0; i < 9; i++) ym.l[i] = fq9_lent_kp<3>(i) - y.l[i];
            fq9_hook("ym", ym);
            CHECK(fq9_to_fq(fmul9(ym, Z)) == fmul(fneg(fq9_to_fq(y)), zc), "lazy 3p - y times z");
            Fq9 x3 = y; // reuse the pattern for X3 < 8p: scale the top limb
            x3.l[8] = (it % 5 == 0) ? Fq9C::KP8[8] : (uint32_t)(rnd() % (Fq9C::KP8[8] + 1u));
            if (!below_kp(x3, 8)) { for (int i = 0; i < 8; i++) x3.l[i] = 0; }
            Fq9 q = fmul9(Z, Z), D;
            for (int i = 0; i < 9; i++) D.l[i] = q.l[i] + fq9_lent_kp<10>(i) - x3.l[i];
            fq9_hook("D", D);
            CHECK(fq9_to_fq(fmul9(D, Z)) == fmul(fsub(fq9_to_fq(q), fq9_to_fq(x3)), zc), "lazy q - x3 + 10p times z");
            // X3 = rr + 6p - ppp - 2q at ppp + 2q close to 6p and at 0
            Fq9 rr = fmul9(Z, y), ppp = (it & 4) ? fq9_zero() : y /* < 2p */, qq = (it & 8) ? fq9_zero() : y;
            Fq9 r3 = fq9_x3(rr, ppp, qq);
            CHECK(normalised(r3) && below_kp(r3, 8), "fq9_x3 bound");
            CHECK(fq9_to_fq(r3) == fsub(fsub(fq9_to_fq(rr), fq9_to_fq(ppp)), fdbl(fq9_to_fq(qq))), "fq9_x3 value");
        }
        CHECK(hook_calls > 10000, "acc9 hook saw the lazy values");
    }
    // ---- Fr on the same representation (NTT chain): values, the < 2r invariant, conversions
    {
        auto rand_fr = [&]() { Fr x; for (int i = 0; i < 8; i += 2) { uint64_t r = rnd(); x.v[i] = (uint32_t)r; x.v[i + 1] = (uint32_t)(r >> 32); } x.v[7] &= 0x0fffffff; return x; };
        auto below_kr = [&](const Fr9& v, int K) {
            uint64_t kp[9], c = 0;
            for (int i = 0; i < 9; i++) { uint64_t t = (uint64_t)Fr9C::P[i] * K + c; if (i < 8) { kp[i] = t & Fr9C::MASK; c = t >> 29; } else kp[i] = t; }
            for (int i = 8; i >= 0; i--) if (v.l[i] != kp[i]) return v.l[i] < kp[i];
            return false;
        };
        for (int it = 0; it < 20000; it++) {
            Fr a = rand_fr(), b = rand_fr();
            if (it == 0) a = Fr::zero();
            if (it == 1) a = Fr::one();
            if (it == 2) { for (int i = 0; i < 8; i++) a.v[i] = FrParams::P[i]; a.v[0] -= 1; b = a; }
            Fr9 A = fr9_from_fr(a), B = fr9_from_fr(b);
            CHECK(fr9_to_fr(A) == a, "fr roundtrip");
            CHECK(fr9_to_fr(frmul9(A, B)) == fmul(a, b), "fr mul");
            Fr9 s = fradd9(A, B), d = frsub9(A, B);
            CHECK(normalised(s) && below_kr(s, 3) && normalised(d) && below_kr(d, 3), "fr invariant");
            CHECK(fr9_to_fr(s) == fadd(a, b) && fr9_to_fr(d) == fsub(a, b), "fr add/sub");
            CHECK(fr9_to_standard(A) == from_mont(a), "fr to standard form");
            // packed round trip (what the kernels keep in HBM)
            uint32_t w[8]; fr9_store(w, s); Fr9 s2 = fr9_load(w);
            bool same = true; for (int i = 0; i < 9; i++) same &= s2.l[i] == s.l[i];
            CHECK(same, "fr pack/unpack");
            // a butterfly chain keeps the invariant
            Fr9 u = A, tt = B; Fr uc = a, tc = b;
            for (int k = 0; k < 24; k++) { Fr9 m = frmul9(tt, B); Fr mc = fmul(tc, b); Fr9 nu = fradd9(u, m); tt = frsub9(u, m); u = nu; Fr nuc = fadd(uc, mc); tc = fsub(uc, mc); uc = nuc;
                CHECK(below_kr(u, 3) && below_kr(tt, 3), "fr butterfly invariant"); }
            CHECK(fr9_to_fr(u) == uc && fr9_to_fr(tt) == tc, "fr butterfly chain");
            // the lazy butterflies of an NTT pass: no reduction for up to 12 stages (bound 2 + 2t), one fred9 at the store
            {
                Fr9 lu = A, lt = B; Fr luc = a, ltc = b;
                for (int k = 1; k <= 12; k++) {
                    Fr9 m = frmul9(B, lt); Fr mc = fmul(b, ltc);
                    Fr9 nu = fadd9(lu, m); lt = fsub9_t<Fr9C, 2>(lu, m); lu = nu;
                    Fr nuc = fadd(luc, mc); ltc = fsub(luc, mc); luc = nuc;
                    CHECK(normalised(lu) && normalised(lt) && below_kr(lu, 2 + 2 * k) && below_kr(lt, 2 + 2 * k), "lazy butterfly bound");
                }
                Fr9 ru = fred9_t<Fr9C>(lu), rt = fred9_t<Fr9C>(lt);
                CHECK(below_kr(ru, 2) && below_kr(rt, 2) && normalised(ru) && normalised(rt), "fred9 after a lazy pass");
                CHECK(fr9_to_fr(ru) == luc && fr9_to_fr(rt) == ltc && fr9_to_fr(lu) == luc && fr9_to_fr(lt) == ltc, "lazy butterfly values");
                uint32_t pw[8]; fr9_store(pw, ru); Fr9 back = fr9_load(pw);
                CHECK(fr9_to_fr(back) == luc, "lazy pass result survives packing");
            }
        }
    }
    // ---- Fq2 over Fq9 and the generic XYZZ formulas instantiated on it (G2)
    {
        auto rand_fq2 = [&]() { return Fq2{rand_fq(), rand_fq()}; };
        auto ok2 = [&](const Fq2n& v) { return normalised(v.a) && normalised(v.b) && below_kp(v.a, 3) && below_kp(v.b, 3); };
        for (int it = 0; it < 5000; it++) {
            Fq2 a = rand_fq2(), b = rand_fq2();
            if (it == 0) a = Fq2::zero();
            if (it == 1) b = Fq2::one();
            if (it == 2) { for (int i = 0; i < 8; i++) { a.a.v[i] = FqParams::P[i]; a.b.v[i] = FqParams::P[i]; } a.a.v[0] -= 1; a.b.v[0] -= 2; b = a; }
            Fq2n A = fq2n_from_canonical(a), B = fq2n_from_canonical(b);
            Fq2n m = fmul(A, B), sq = fsqr(A), ad = fadd(A, B), sb = fsub(A, B), ng = fneg(A), db = fdbl(A);
            CHECK(ok2(m) && ok2(sq) && ok2(ad) && ok2(sb) && ok2(ng) && ok2(db), "fq2n invariant");
            Fq2 cm = fq2n_to_canonical(m), wm = fmul(a, b);
            CHECK(cm.a == wm.a && cm.b == wm.b, "fq2n mul");
            Fq2 cs = fq2n_to_canonical(sq), ws = fsqr(a);
            CHECK(cs.a == ws.a && cs.b == ws.b, "fq2n sqr");
            Fq2 ca = fq2n_to_canonical(ad), wa = fadd(a, b);
            CHECK(ca.a == wa.a && ca.b == wa.b, "fq2n add");
            Fq2 cb = fq2n_to_canonical(sb), wb = fsub(a, b);
            CHECK(cb.a == wb.a && cb.b == wb.b, "fq2n sub");
            // chained: invariant must survive many operations
            Fq2n x = A; Fq2 xc = a;
            for (int k = 0; k < 20; k++) { x = fsub(fmul(fadd(x, B), x), fdbl(B)); xc = fsub(fmul(fadd(xc, b), xc), fdbl(b)); CHECK(ok2(x), "fq2n chain invariant"); }
            Fq2 xx = fq2n_to_canonical(x);
            CHECK(xx.a == xc.a && xx.b == xc.b, "fq2n chain value");
            CHECK(fsub(A, A).is_zero() && !fadd(A, Fq2n::one()).is_zero() == !(fadd(a, Fq2::one()).is_zero()), "fq2n is_zero");
        }
        // G2 generator and a table
        static const char* const G2S[4] = {
            "10857046999023057135944570762232829481370756359578518086990519993285655852781",
            "11559732032986387107991004021392285783925812861821192530917403151452391805634",
            "8495653923123431417604973247489272438418190587263600148770280649306958101930",
            "4082367875863433681332203403145435568316851327593401208105741076214120093531"};
        Fq v[4];
        Fq ten = Fq::zero(); ten.v[0] = 10; ten = to_mont(ten);
        for (int k = 0; k < 4; k++) {
            Fq acc = Fq::zero();
            for (const char* p = G2S[k]; *p; p++) { Fq d = Fq::zero(); d.v[0] = (uint32_t)(*p - '0'); acc = fadd(fmul(acc, ten), to_mont(d)); }
            v[k] = acc;
        }
        G2Aff g2{Fq2{v[0], v[1]}, Fq2{v[2], v[3]}};
        const int N2 = 24;
        G2Aff tb[N2]; Aff<Fq2n> tb9[N2];
        G2Xyzz acc = G2Xyzz::zero();
        for (int i = 0; i < N2; i++) {
            acc = padd_mixed(acc, g2);
            tb[i] = to_affine(acc);
            if (i == 5) tb[i] = G2Aff{Fq2::zero(), Fq2::zero()};
            tb9[i] = Aff<Fq2n>{fq2n_from_canonical(tb[i].x), fq2n_from_canonical(tb[i].y)};
        }
        auto same2 = [&](const Xyzz<Fq2n>& a, const G2Xyzz& b) {
            if (b.is_zero()) return a.is_zero();
            Fq2 x = fq2n_to_canonical(a.x), y = fq2n_to_canonical(a.y), zz = fq2n_to_canonical(a.zz), zzz = fq2n_to_canonical(a.zzz);
            return x.a == b.x.a && x.b == b.x.b && y.a == b.y.a && y.b == b.y.b && zz.a == b.zz.a && zz.b == b.zz.b && zzz.a == b.zzz.a && zzz.b == b.zzz.b;
        };
        for (int walk = 0; walk < 60; walk++) {
            G2Xyzz c = G2Xyzz::zero(), c2 = G2Xyzz::zero();
            Xyzz<Fq2n> n = Xyzz<Fq2n>::zero(), n2 = Xyzz<Fq2n>::zero();
            for (int step = 0; step < 40; step++) {
                int op = (int)(rnd() % 7), k = (int)(rnd() % N2);
                if (step == 1 || step == 2) { op = 0; k = 3; }
                if (op <= 3) { c = padd_mixed(c, tb[k]); n = padd_mixed(n, tb9[k]); }
                else if (op == 4) { c = pdbl(c); n = pdbl(n); }
                else if (op == 5) { c2 = padd(c2, c); n2 = padd(n2, n); }
                else { c = padd(c, c2); n = padd(n, n2); }
                CHECK(ok2(n.x) && ok2(n.y) && ok2(n.zz) && ok2(n.zzz), "G2 point invariant");
                CHECK(same2(n, c) && same2(n2, c2), "G2 XYZZ representation equal");
            }
            // P + (-P) through the mixed add
            G2Xyzz cc = G2Xyzz::from_aff(tb[walk % N2 == 5 ? 6 : walk % N2]);
            Xyzz<Fq2n> nn = Xyzz<Fq2n>::from_aff(tb9[walk % N2 == 5 ? 6 : walk % N2]);
            G2Aff ng = pneg(tb[walk % N2 == 5 ? 6 : walk % N2]);
            Aff<Fq2n> ng9{fq2n_from_canonical(ng.x), fq2n_from_canonical(ng.y)};
            CHECK(padd_mixed(nn, ng9).is_zero() && padd_mixed(cc, ng).is_zero(), "G2 madd P+(-P)");
            CHECK(same2(padd_mixed(nn, tb9[walk % N2 == 5 ? 6 : walk % N2]), padd_mixed(cc, tb[walk % N2 == 5 ? 6 : walk % N2])), "G2 madd P+P");
        }
    }
    printf(fails ? "FAILED %d checks\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
