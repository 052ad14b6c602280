// bn254_pairing.h -- optimal-ate pairing on BN254 and the Groth16 pairing check, for the batched verifier
// (include/k16.h: k16_verify_batch).  __host__ __device__: the kernels of verify.hip and the host-side cross-check
// (tests/cpp/pairing_check.cpp) compile the same code.
//
// What it replaces.  The service re-verifies every proof on the CPU before releasing it
// (prover-service/src/request_handler/prover_handler.rs:329-336 -> aptos-types Groth16Proof::verify_proof ->
// ark-groth16 0.4.0 verify_proof_with_prepared_inputs -> ark-ec 0.4.2 Bn::multi_miller_loop / final_exponentiation on
// ark-bn254 0.4.0 -- third-party crates pinned in Cargo.lock:501-639, not vendored under /root/reference).  The pairing
// VALUE is fixed by the exponent of the final exponentiation, which follows ark-ec's (easy part, then the
// Fuentes-Castaneda hard part, i.e. (p^12 - 1)/r times 2x(6x^2 + 3x + 1)); how the tower arithmetic is organised does not
// change any value (field elements are canonical), so it is organised for the GPU here:
//   Fp6 by Karatsuba (6 Fq2 products), Fp12 squaring by the complex method, sparse line multiplication (13 Fq2 products),
//   Granger-Scott squaring in the cyclotomic subgroup for the three x-powers of the hard part.
// One lane computes one Miller loop or one final exponentiation: a batch of proofs is 3n independent Miller loops and n
// final exponentiations.  The large functions are deliberately NOT inlined (one copy of each in the code object).
#pragma once
#include "bn254_curve.h"

#define K16_HDN __host__ __device__ __attribute__((noinline))

namespace k16 {

struct Fp6 {
    Fq2 c0, c1, c2; // c0 + c1 v + c2 v^2, v^3 = xi = 9 + u
};
struct Fp12 {
    Fp6 c0, c1; // c0 + c1 w, w^2 = v
};
// curve / tower constants in Montgomery form, filled once on the host (pairing_consts_init) and handed to the kernels
struct PairConsts {
    Fq2 twist_b;        // 3 / (9 + u)
    Fq2 twqx, twqy;     // xi^((p-1)/3), xi^((p-1)/2): the Frobenius on the twist
    Fq2 frob6_c1[4];    // xi^((p^k - 1)/3), k = 1..3
    Fq2 frob6_c2[4];    // xi^(2 (p^k - 1)/3)
    Fq2 frob12_c1[4];   // xi^((p^k - 1)/6)
    Fq  two_inv;
};

// ---------------------------------------------------------------- Fq2 helpers
K16_HD Fq2 fconj(const Fq2& x) { return Fq2{x.a, fneg(x.b)}; }
K16_HD Fq2 fmul_fp(const Fq2& x, const Fq& k) { return Fq2{fmul(x.a, k), fmul(x.b, k)}; }
// (a + b u)(9 + u) = (9a - b) + (a + 9b) u, by additions only
K16_HD Fq2 fmul_xi(const Fq2& x)
{
    Fq a8 = fdbl(fdbl(fdbl(x.a))), b8 = fdbl(fdbl(fdbl(x.b)));
    return Fq2{fsub(fadd(a8, x.a), x.b), fadd(fadd(b8, x.b), x.a)};
}

// ---------------------------------------------------------------- Fp6
K16_HD Fp6 f6_zero() { return Fp6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
K16_HD Fp6 f6_add(const Fp6& x, const Fp6& y) { return Fp6{fadd(x.c0, y.c0), fadd(x.c1, y.c1), fadd(x.c2, y.c2)}; }
K16_HD Fp6 f6_sub(const Fp6& x, const Fp6& y) { return Fp6{fsub(x.c0, y.c0), fsub(x.c1, y.c1), fsub(x.c2, y.c2)}; }
K16_HD Fp6 f6_neg(const Fp6& x) { return Fp6{fneg(x.c0), fneg(x.c1), fneg(x.c2)}; }
K16_HD Fp6 f6_mul_v(const Fp6& x) { return Fp6{fmul_xi(x.c2), x.c0, x.c1}; }
// Karatsuba over Fq2: 6 products
K16_HDN void f6_mul(Fp6* r, const Fp6* x, const Fp6* y)
{
    Fq2 v0 = fmul(x->c0, y->c0), v1 = fmul(x->c1, y->c1), v2 = fmul(x->c2, y->c2);
    Fq2 t0 = fsub(fsub(fmul(fadd(x->c1, x->c2), fadd(y->c1, y->c2)), v1), v2); // x1 y2 + x2 y1
    Fq2 t1 = fsub(fsub(fmul(fadd(x->c0, x->c1), fadd(y->c0, y->c1)), v0), v1); // x0 y1 + x1 y0
    Fq2 t2 = fsub(fsub(fmul(fadd(x->c0, x->c2), fadd(y->c0, y->c2)), v0), v2); // x0 y2 + x2 y0
    r->c0  = fadd(v0, fmul_xi(t0));
    r->c1  = fadd(t1, fmul_xi(v2));
    r->c2  = fadd(t2, v1);
}
// (b0 + b1 v) * x : 5 products
K16_HDN void f6_mul_by_01(Fp6* r, const Fp6* x, const Fq2* b0, const Fq2* b1)
{
    Fq2 v0 = fmul(x->c0, *b0), v1 = fmul(x->c1, *b1);
    Fq2 t1 = fsub(fsub(fmul(fadd(x->c0, x->c1), fadd(*b0, *b1)), v0), v1); // x0 b1 + x1 b0
    Fq2 x2b0 = fmul(x->c2, *b0), x2b1 = fmul(x->c2, *b1);
    r->c0 = fadd(v0, fmul_xi(x2b1));
    r->c1 = t1;
    r->c2 = fadd(x2b0, v1);
}
K16_HDN void f6_inv(Fp6* r, const Fp6* x)
{
    Fq2 s0 = fsub(fsqr(x->c0), fmul_xi(fmul(x->c1, x->c2)));
    Fq2 s1 = fsub(fmul_xi(fsqr(x->c2)), fmul(x->c0, x->c1));
    Fq2 s2 = fsub(fsqr(x->c1), fmul(x->c0, x->c2));
    Fq2 n  = fadd(fmul(x->c0, s0), fmul_xi(fadd(fmul(x->c2, s1), fmul(x->c1, s2))));
    Fq2 ni = finv(n);
    r->c0  = fmul(s0, ni);
    r->c1  = fmul(s1, ni);
    r->c2  = fmul(s2, ni);
}
K16_HD Fp6 f6_frob(const Fp6& x, int k, const PairConsts& K)
{
    Fq2 c0 = x.c0, c1 = x.c1, c2 = x.c2;
    if (k & 1) {
        c0 = fconj(c0);
        c1 = fconj(c1);
        c2 = fconj(c2);
    }
    return Fp6{c0, fmul(c1, K.frob6_c1[k]), fmul(c2, K.frob6_c2[k])};
}

// ---------------------------------------------------------------- Fp12
K16_HD Fp12 f12_one() { return Fp12{Fp6{Fq2::one(), Fq2::zero(), Fq2::zero()}, f6_zero()}; }
K16_HD bool f12_eq(const Fp12& x, const Fp12& y)
{
    return x.c0.c0 == y.c0.c0 && x.c0.c1 == y.c0.c1 && x.c0.c2 == y.c0.c2 && x.c1.c0 == y.c1.c0 && x.c1.c1 == y.c1.c1 &&
           x.c1.c2 == y.c1.c2;
}
K16_HD Fp12 f12_conj(const Fp12& x) { return Fp12{x.c0, f6_neg(x.c1)}; } // x^(p^6): the inverse of a unitary element
K16_HDN void f12_mul(Fp12* r, const Fp12* x, const Fp12* y)
{
    Fp6 v0, v1, t, s0 = f6_add(x->c0, x->c1), s1 = f6_add(y->c0, y->c1);
    f6_mul(&v0, &x->c0, &y->c0);
    f6_mul(&v1, &x->c1, &y->c1);
    f6_mul(&t, &s0, &s1);
    r->c1 = f6_sub(f6_sub(t, v0), v1);
    r->c0 = f6_add(v0, f6_mul_v(v1));
}
// complex squaring: c0 = (a0 + a1)(a0 + v a1) - ab - v ab, c1 = 2ab
K16_HDN void f12_sqr(Fp12* r, const Fp12* x)
{
    Fp6 ab, t, s0 = f6_add(x->c0, x->c1), s1 = f6_add(x->c0, f6_mul_v(x->c1));
    f6_mul(&ab, &x->c0, &x->c1);
    f6_mul(&t, &s0, &s1);
    r->c0 = f6_sub(f6_sub(t, ab), f6_mul_v(ab));
    r->c1 = f6_add(ab, ab);
}
K16_HDN void f12_inv(Fp12* r, const Fp12* x)
{
    Fp6 t0, t1, d;
    f6_mul(&t0, &x->c0, &x->c0);
    f6_mul(&t1, &x->c1, &x->c1);
    d = f6_sub(t0, f6_mul_v(t1));
    f6_inv(&t1, &d);
    f6_mul(&r->c0, &x->c0, &t1);
    f6_mul(&t0, &x->c1, &t1);
    r->c1 = f6_neg(t0);
}
K16_HDN void f12_frob(Fp12* r, const Fp12* x, int k, const PairConsts* K)
{
    Fp6 c0 = f6_frob(x->c0, k, *K), c1 = f6_frob(x->c1, k, *K);
    r->c0  = c0;
    r->c1  = Fp6{fmul(c1.c0, K->frob12_c1[k]), fmul(c1.c1, K->frob12_c1[k]), fmul(c1.c2, K->frob12_c1[k])};
}
// f * (c0 + (d0 + d1 v) w): the value of a line at P (D-type twist), 13 Fq2 products instead of 18
K16_HDN void f12_mul_by_034(Fp12* f, const Fq2* c0, const Fq2* d0, const Fq2* d1)
{
    Fp6 a{fmul(f->c0.c0, *c0), fmul(f->c0.c1, *c0), fmul(f->c0.c2, *c0)};
    Fp6 b, e, s = f6_add(f->c0, f->c1);
    f6_mul_by_01(&b, &f->c1, d0, d1);
    Fq2 cd = fadd(*c0, *d0);
    f6_mul_by_01(&e, &s, &cd, d1);
    f->c1 = f6_sub(e, f6_add(a, b));
    f->c0 = f6_add(f6_mul_v(b), a);
}
// Granger-Scott squaring, valid in the cyclotomic subgroup (after the easy part of the final exponentiation):
// 9 Fq2 products instead of 12.  With f = g0 + g1 w, g_i = (g_i0, g_i1, g_i2): the three Fp4 squarings
// (g00, g11), (g10, g02), (g01, g12).
K16_HDN void f12_cyclo_sqr(Fp12* r, const Fp12* x)
{
    auto fp4_sqr = [](const Fq2& a, const Fq2& b, Fq2& t0, Fq2& t1) { // (a + b y)^2, y^2 = xi
        Fq2 ab = fmul(a, b);
        t0 = fsub(fsub(fmul(fadd(a, b), fadd(fmul_xi(b), a)), ab), fmul_xi(ab));
        t1 = fdbl(ab);
    };
    Fq2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(x->c0.c0, x->c1.c1, t0, t1);
    fp4_sqr(x->c1.c0, x->c0.c2, t2, t3);
    fp4_sqr(x->c0.c1, x->c1.c2, t4, t5);
    auto m3p2 = [](const Fq2& t, const Fq2& z) { return fadd(fdbl(fadd(t, z)), t); }; // 3t + 2z
    auto m3m2 = [](const Fq2& t, const Fq2& z) { return fadd(fdbl(fsub(t, z)), t); }; // 3t - 2z
    Fp12 o;
    o.c0.c0 = m3m2(t0, x->c0.c0);
    o.c1.c1 = m3p2(t1, x->c1.c1);
    o.c1.c0 = m3p2(fmul_xi(t5), x->c1.c0);
    o.c0.c2 = m3m2(t4, x->c0.c2);
    o.c0.c1 = m3m2(t2, x->c0.c1);
    o.c1.c2 = m3p2(t3, x->c1.c2);
    *r = o;
}
// f^(-x), x = 4965661367192848881 (ark-bn254 Config::X, positive): cyclotomic square-and-multiply, then the conjugate
K16_HDN void f12_exp_by_neg_x(Fp12* r, const Fp12* f)
{
    const uint64_t X = 4965661367192848881ull;
    Fp12           acc = *f; // top bit of X (bit 62)
#pragma clang loop unroll(disable)
    for (int i = 61; i >= 0; i--) {
        f12_cyclo_sqr(&acc, &acc);
        if ((X >> i) & 1) f12_mul(&acc, &acc, f);
    }
    *r = f12_conj(acc);
}

// ---------------------------------------------------------------- Miller loop (homogeneous projective G2, D-type twist)
struct G2Hom {
    Fq2 x, y, z;
};
struct Ell {
    Fq2 c0, c1, c2;
};
K16_HDN void g2hom_double(G2Hom* r, Ell* l, const PairConsts* K)
{
    Fq2 a = fmul_fp(fmul(r->x, r->y), K->two_inv);
    Fq2 b = fsqr(r->y), c = fsqr(r->z);
    Fq2 e = fmul(K->twist_b, fadd(fdbl(c), c));
    Fq2 f = fadd(fdbl(e), e);
    Fq2 g = fmul_fp(fadd(b, f), K->two_inv);
    Fq2 h = fsub(fsqr(fadd(r->y, r->z)), fadd(b, c));
    Fq2 i = fsub(e, b);
    Fq2 j = fsqr(r->x);
    Fq2 e2 = fsqr(e);
    r->x  = fmul(a, fsub(b, f));
    r->y  = fsub(fsqr(g), fadd(fdbl(e2), e2));
    r->z  = fmul(b, h);
    l->c0 = fneg(h);
    l->c1 = fadd(fdbl(j), j);
    l->c2 = i;
}
K16_HDN void g2hom_add(G2Hom* r, const Aff<Fq2>* q, Ell* l)
{
    Fq2 theta  = fsub(r->y, fmul(q->y, r->z));
    Fq2 lambda = fsub(r->x, fmul(q->x, r->z));
    Fq2 c = fsqr(theta), d = fsqr(lambda);
    Fq2 e = fmul(lambda, d), f = fmul(r->z, c), g = fmul(r->x, d);
    Fq2 h = fsub(fadd(e, f), fdbl(g));
    Fq2 ny = fsub(fmul(theta, fsub(g, h)), fmul(e, r->y));
    r->x  = fmul(lambda, h);
    r->y  = ny;
    r->z  = fmul(r->z, e);
    l->c0 = lambda;
    l->c1 = fneg(theta);
    l->c2 = fsub(fmul(theta, q->x), fmul(lambda, q->y));
}
K16_HD void f12_ell(Fp12* f, const Ell& l, const Aff<Fq>& p)
{
    Fq2 c0 = fmul_fp(l.c0, p.y), c1 = fmul_fp(l.c1, p.x);
    f12_mul_by_034(f, &c0, &c1, &l.c2);
}
K16_HD Aff<Fq2> g2_mul_by_char(const Aff<Fq2>& q, const PairConsts& K)
{
    return Aff<Fq2>{fmul(fconj(q.x), K.twqx), fmul(fconj(q.y), K.twqy)};
}
// signed digits of 6x + 2, least significant first (ark-bn254 Config::ATE_LOOP_COUNT), packed: bit i of NZ = digit i is
// non-zero, bit i of NEG = it is -1
constexpr uint64_t ATE_NZ_LO  = 0xa5899049c2964ca8ull; // digits 0..63  (computed from the digit list, checked in tests)
constexpr uint64_t ATE_NEG_LO = 0x0408100802100880ull;
constexpr unsigned ATE_TOP    = 64;                    // digit 64 = +1 (the leading digit)

// Miller loop of one pair; a zero P or Q contributes 1 (ark-ec multi_miller_loop filters such pairs out)
K16_HDN void miller_loop(Fp12* out, const Aff<Fq>* p, const Aff<Fq2>* q, const PairConsts* K)
{
    Fp12 f = f12_one();
    if (p->is_zero() || q->is_zero()) {
        *out = f;
        return;
    }
    G2Hom    r{q->x, q->y, Fq2::one()};
    Aff<Fq2> nq{q->x, fneg(q->y)};
    Ell      l;
#pragma clang loop unroll(disable)
    for (int i = (int)ATE_TOP; i >= 1; i--) {
        if (i != (int)ATE_TOP) f12_sqr(&f, &f);
        g2hom_double(&r, &l, K);
        f12_ell(&f, l, *p);
        const unsigned d = (unsigned)(i - 1);
        if ((ATE_NZ_LO >> d) & 1) {
            g2hom_add(&r, ((ATE_NEG_LO >> d) & 1) ? &nq : q, &l);
            f12_ell(&f, l, *p);
        }
    }
    Aff<Fq2> q1 = g2_mul_by_char(*q, *K);
    Aff<Fq2> q2 = g2_mul_by_char(q1, *K);
    q2.y        = fneg(q2.y);
    g2hom_add(&r, &q1, &l);
    f12_ell(&f, l, *p);
    g2hom_add(&r, &q2, &l);
    f12_ell(&f, l, *p);
    *out = f;
}

// ark-ec models/bn/mod.rs final_exponentiation.  Returns false for f = 0 (no inverse; never a Miller-loop output).
K16_HDN bool final_exponentiation(Fp12* out, const Fp12* f, const PairConsts* K)
{
    {
        Fp12 z{f6_zero(), f6_zero()};
        if (f12_eq(*f, z)) return false;
    }
    Fp12 f1 = f12_conj(*f), f2, r, y0, y1, y2, y3, y4, y5, y6, t;
    f12_inv(&f2, f);
    f12_mul(&r, &f1, &f2);       // f^(p^6 - 1)
    f2 = r;
    f12_frob(&r, &r, 2, K);
    f12_mul(&r, &r, &f2);        // f^((p^6 - 1)(p^2 + 1)): unitary from here on
    f12_exp_by_neg_x(&y0, &r);
    f12_cyclo_sqr(&y1, &y0);
    f12_cyclo_sqr(&y2, &y1);
    f12_mul(&y3, &y2, &y1);
    f12_exp_by_neg_x(&y4, &y3);
    f12_cyclo_sqr(&y5, &y4);
    f12_exp_by_neg_x(&y6, &y5);
    y3 = f12_conj(y3);
    y6 = f12_conj(y6);
    Fp12 y7, y8, y9, y10, y11, y12, y13, y14, y15;
    f12_mul(&y7, &y6, &y4);
    f12_mul(&y8, &y7, &y3);
    f12_mul(&y9, &y8, &y1);
    f12_mul(&y10, &y8, &y4);
    f12_mul(&y11, &y10, &r);
    f12_frob(&y12, &y9, 1, K);
    f12_mul(&y13, &y12, &y11);
    f12_frob(&t, &y8, 2, K);
    f12_mul(&y14, &t, &y13);
    r = f12_conj(r);
    f12_mul(&y15, &r, &y9);
    f12_frob(&t, &y15, 3, K);
    f12_mul(out, &t, &y14);
    return true;
}

// ---------------------------------------------------------------- host: constants
// decimal (standard form) -> Montgomery, on the host
inline Fq fq_from_dec(const char* s)
{
    Fq ten = Fq::zero();
    ten.v[0] = 10;
    ten      = to_mont(ten);
    Fq acc   = Fq::zero();
    for (const char* p = s; *p; p++) {
        Fq d   = Fq::zero();
        d.v[0] = (uint32_t)(*p - '0');
        acc    = fadd(fmul(acc, ten), to_mont(d));
    }
    return acc;
}
inline void pairing_consts_init(PairConsts* K)
{
    auto f2 = [](const char* a, const char* b) { return Fq2{fq_from_dec(a), fq_from_dec(b)}; };
    // values: xi^((p^k-1)/3), xi^(2(p^k-1)/3), xi^((p^k-1)/6) for xi = 9 + u (ark-bn254 fq6.rs / fq12.rs FROBENIUS_COEFF tables),
    // 3/xi (g2.rs COEFF_B), xi^((p-1)/3), xi^((p-1)/2) (TWIST_MUL_BY_Q_X / _Y); recomputed by tests/test_oracle_pairing.py
    K->twist_b = f2("19485874751759354771024239261021720505790618469301721065564631296452457478373",
                    "266929791119991161246907387137283842545076965332900288569378510910307636690");
    K->twqx = f2("21575463638280843010398324269430826099269044274347216827212613867836435027261",
                 "10307601595873709700152284273816112264069230130616436755625194854815875713954");
    K->twqy = f2("2821565182194536844548159561693502659359617185244120367078079554186484126554",
                 "3505843767911556378687030309984248845540243509899259641013678093033130930403");
    K->frob6_c1[0] = K->frob6_c2[0] = K->frob12_c1[0] = Fq2::one();
    K->frob6_c1[1] = K->twqx;
    K->frob6_c2[1] = f2("2581911344467009335267311115468803099551665605076196740867805258568234346338",
                        "19937756971775647987995932169929341994314640652964949448313374472400716661030");
    K->frob12_c1[1] = f2("8376118865763821496583973867626364092589906065868298776909617916018768340080",
                         "16469823323077808223889137241176536799009286646108169935659301613961712198316");
    K->frob6_c1[2] = f2("21888242871839275220042445260109153167277707414472061641714758635765020556616", "0");
    K->frob6_c2[2] = f2("2203960485148121921418603742825762020974279258880205651966", "0");
    K->frob12_c1[2] = f2("21888242871839275220042445260109153167277707414472061641714758635765020556617", "0");
    K->frob6_c1[3] = f2("3772000881919853776433695186713858239009073593817195771773381919316419345261",
                        "2236595495967245188281701248203181795121068902605861227855261137820944008926");
    K->frob6_c2[3] = f2("5324479202449903542726783395506214481928257762400643279780343368557297135718",
                        "16208900380737693084919495127334387981393726419856888799917914180988844123039");
    K->frob12_c1[3] = f2("11697423496358154304825782922584725312912383441159505038794027105778954184319",
                         "303847389135065887422783454877609941456349188919719272345083954437860409601");
    K->two_inv = fq_from_dec("10944121435919637611123202872628637544348155578648911831344518947322613104292");
}

} // namespace k16
