/*
 * oracle/pairing_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Included by bn254_ref.c.
 *
 * CPU restatement of the proof check the service runs on every proof before releasing it
 * (prover-service/src/request_handler/prover_handler.rs:329-336):
 *     groth16_proof.verify_proof(public_inputs_hash, &prepared_vk)
 * That call lives in third-party crates that are NOT under /root/reference:
 *     aptos-types (aptos-core @ b163f034c0e1e01b59cef44694da9dbccb4d5b63, Cargo.lock:444-446)
 *       -> ark-groth16 0.4.0  Groth16::verify_proof_with_prepared_inputs
 *       -> ark-ec 0.4.2       models::bn::{Bn::multi_miller_loop, Bn::final_exponentiation, g2::G2Prepared}
 *       -> ark-bn254 0.4.0    curve constants (X = 4965661367192848881, D-type twist, xi = 9 + u)
 *       -> ark-ff 0.4.2       Fp2 / Fp6 (3 over 2) / Fp12 (2 over 3 over 2) towers
 * so this file restates their PUBLISHED algorithm (Cargo.lock pins the versions above):
 *   - G2 line coefficients in homogeneous projective coordinates, doubling and mixed addition steps
 *     (ark-ec models/bn/g2.rs: G2HomProjective::double_in_place / add_in_place, mul_by_char)
 *   - Miller loop over the signed digits of 6x + 2, then the two Frobenius additions Q1, -Q2, line
 *     evaluation "ell" = multiplication by a sparse element (models/bn/mod.rs: multi_miller_loop, ell)
 *   - final exponentiation: easy part (p^6 - 1)(p^2 + 1), hard part after Fuentes-Castaneda et al., which raises to
 *     2x(6x^2 + 3x + 1) * (p^4 - p^2 + 1)/r (models/bn/mod.rs: final_exponentiation, exp_by_neg_x)
 *   - Groth16: e(A, B) * e(vk_x, -gamma) * e(C, -delta) == e(alpha, beta),  vk_x = IC[0] + sum_i x_i IC[i+1]
 *     (ark-groth16 src/verifier.rs: prepare_inputs, verify_proof_with_prepared_inputs)
 * The pairing VALUE does not depend on how the tower arithmetic is organised (field elements are canonical), only on
 * the exponent of the final exponentiation, which is reproduced exactly.
 *
 * Pinned by (tests/test_oracle_pairing.py): the reference's own acceptance test -- the toy circuit's proof verifies
 * under prover-service/resources/toy_circuit/toy_vk.json with public input 2 (tests/prover_handler.rs:279-290) and is
 * rejected with 3; bilinearity e(aP, bQ) = e(P, Q)^(ab); and the independent pure-Python pairing of
 * tests/bn254_pairing.py raised to 2x(6x^2+3x+1).
 */
#ifndef ORACLE_PAIRING_REF_H
#define ORACLE_PAIRING_REF_H

typedef struct { fq2_t c0, c1, c2; } fq6_t;   /* c0 + c1 v + c2 v^2, v^3 = xi = 9 + u */
typedef struct { fq6_t c0, c1; } fq12_t;      /* c0 + c1 w, w^2 = v */

static fq2_t PR_XI, PR_TWIST_B, PR_TWQX, PR_TWQY, PR_FROB6_C1[4], PR_FROB6_C2[4], PR_FROB12_C1[4];
static fe_t  PR_TWO_INV;
static int   pr_ready = 0;

static void fq2_from_dec(fq2_t* r, const char* a, const char* b)
{
    fe_from_dec(FQ, &r->a, a);
    fe_from_dec(FQ, &r->b, b);
}
static void fq2_pow_big(fq2_t* r, const fq2_t* base, const u64* e, int nlimbs)
{
    fq2_t acc;
    fq2_one(&acc);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        fq2_sqr(&acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fq2_mul(&acc, &acc, base);
    }
    *r = acc;
}
/* big little-endian integer helpers for the exponents (p^k - 1)/d, k <= 3 */
static void big_mul_small(u64* r, int* n, const u64* a, int na, const u64* b, int nb)
{
    u64 t[16] = {0};
    for (int i = 0; i < na; i++) {
        u128 c = 0;
        for (int j = 0; j < nb; j++) {
            c += (u128)a[i] * b[j] + t[i + j];
            t[i + j] = (u64)c;
            c >>= 64;
        }
        t[i + nb] += (u64)c;
    }
    *n = na + nb;
    memcpy(r, t, sizeof(u64) * (size_t)(*n));
}
static void big_sub1_div(u64* a, int n, u64 d)
{
    for (int i = 0; i < n; i++) {
        if (a[i]-- != 0) break;
    }
    u128 rem = 0;
    for (int i = n - 1; i >= 0; i--) {
        u128 cur = (rem << 64) | a[i];
        a[i]     = (u64)(cur / d);
        rem      = cur % d;
    }
}
static void pr_init(void)
{
    if (pr_ready) return;
    fq2_from_dec(&PR_XI, "9", "1");
    /* ark-bn254 g2.rs: COEFF_B = 3 / (9 + u) */
    fq2_t three;
    fq2_from_dec(&three, "3", "0");
    fq2_t xinv;
    fq2_inv(&xinv, &PR_XI);
    fq2_mul(&PR_TWIST_B, &three, &xinv);
    fe_t two;
    fe_add(FQ, &two, &ORA_FQ.one, &ORA_FQ.one);
    fe_inv(FQ, &PR_TWO_INV, &two);
    /* Frobenius coefficients xi^((p^k - 1)/3), xi^(2(p^k - 1)/3), xi^((p^k - 1)/6)  (ark-bn254 fq6.rs / fq12.rs tables) */
    u64 pk[16];
    int n = 4;
    memcpy(pk, ORA_FQ.p, 32);
    for (int k = 1; k <= 3; k++) {
        u64 e[16];
        memcpy(e, pk, sizeof(u64) * (size_t)n);
        big_sub1_div(e, n, 3);
        fq2_pow_big(&PR_FROB6_C1[k], &PR_XI, e, n);
        fq2_sqr(&PR_FROB6_C2[k], &PR_FROB6_C1[k]);
        memcpy(e, pk, sizeof(u64) * (size_t)n);
        big_sub1_div(e, n, 6);
        fq2_pow_big(&PR_FROB12_C1[k], &PR_XI, e, n);
        if (k < 3) {
            u64 t[16];
            int nn;
            big_mul_small(t, &nn, pk, n, ORA_FQ.p, 4);
            memcpy(pk, t, sizeof(u64) * (size_t)nn);
            n = nn;
        }
    }
    /* ark-bn254 TWIST_MUL_BY_Q_X = xi^((p-1)/3), TWIST_MUL_BY_Q_Y = xi^((p-1)/2) */
    PR_TWQX = PR_FROB6_C1[1];
    {
        u64 e[4];
        memcpy(e, ORA_FQ.p, 32);
        big_sub1_div(e, 4, 2);
        fq2_pow_big(&PR_TWQY, &PR_XI, e, 4);
    }
    pr_ready = 1;
}

/* ---- Fq2 helpers */
static inline void fq2_mul_fe(fq2_t* r, const fq2_t* x, const fe_t* k)
{
    fe_mul(FQ, &r->a, &x->a, k);
    fe_mul(FQ, &r->b, &x->b, k);
}
static inline void fq2_conj(fq2_t* r, const fq2_t* x)
{
    r->a = x->a;
    fe_neg(FQ, &r->b, &x->b);
}
static inline void fq2_dbl(fq2_t* r, const fq2_t* x) { fq2_add(r, x, x); }
static inline void fq2_mul_xi(fq2_t* r, const fq2_t* x) { fq2_mul(r, x, &PR_XI); }

/* ---- Fq6 = Fq2[v]/(v^3 - xi)  (ark-ff fp6_3over2.rs) */
static void fq6_add(fq6_t* r, const fq6_t* x, const fq6_t* y)
{
    fq2_add(&r->c0, &x->c0, &y->c0);
    fq2_add(&r->c1, &x->c1, &y->c1);
    fq2_add(&r->c2, &x->c2, &y->c2);
}
static void fq6_sub(fq6_t* r, const fq6_t* x, const fq6_t* y)
{
    fq2_sub(&r->c0, &x->c0, &y->c0);
    fq2_sub(&r->c1, &x->c1, &y->c1);
    fq2_sub(&r->c2, &x->c2, &y->c2);
}
static void fq6_neg(fq6_t* r, const fq6_t* x)
{
    fq2_neg(&r->c0, &x->c0);
    fq2_neg(&r->c1, &x->c1);
    fq2_neg(&r->c2, &x->c2);
}
/* schoolbook product reduced with v^3 = xi */
static void fq6_mul(fq6_t* r, const fq6_t* x, const fq6_t* y)
{
    fq2_t t, d0, d1, d2, d3, d4;
    fq2_mul(&d0, &x->c0, &y->c0);
    fq2_mul(&d1, &x->c0, &y->c1);
    fq2_mul(&t, &x->c1, &y->c0);
    fq2_add(&d1, &d1, &t);
    fq2_mul(&d2, &x->c0, &y->c2);
    fq2_mul(&t, &x->c1, &y->c1);
    fq2_add(&d2, &d2, &t);
    fq2_mul(&t, &x->c2, &y->c0);
    fq2_add(&d2, &d2, &t);
    fq2_mul(&d3, &x->c1, &y->c2);
    fq2_mul(&t, &x->c2, &y->c1);
    fq2_add(&d3, &d3, &t);
    fq2_mul(&d4, &x->c2, &y->c2);
    fq2_mul_xi(&d3, &d3);
    fq2_mul_xi(&d4, &d4);
    fq2_add(&r->c0, &d0, &d3);
    fq2_add(&r->c1, &d1, &d4);
    r->c2 = d2;
}
/* multiplication by v: (c0, c1, c2) -> (xi c2, c0, c1) */
static void fq6_mul_v(fq6_t* r, const fq6_t* x)
{
    fq2_t t;
    fq2_mul_xi(&t, &x->c2);
    r->c2 = x->c1;
    r->c1 = x->c0;
    r->c0 = t;
}
static void fq6_inv(fq6_t* r, const fq6_t* x)
{
    /* ark-ff fp6_3over2.rs inverse: "High-Speed Software Implementation of the Optimal Ate Pairing over BN curves", Alg. 17 */
    fq2_t t0, t1, t2, t3, t4, t5, s0, s1, s2, a1, a3;
    fq2_sqr(&t0, &x->c0);
    fq2_sqr(&t1, &x->c1);
    fq2_sqr(&t2, &x->c2);
    fq2_mul(&t3, &x->c0, &x->c1);
    fq2_mul(&t4, &x->c0, &x->c2);
    fq2_mul(&t5, &x->c1, &x->c2);
    fq2_mul_xi(&s0, &t5);
    fq2_sub(&s0, &t0, &s0); /* c0^2 - xi c1 c2 */
    fq2_mul_xi(&s1, &t2);
    fq2_sub(&s1, &s1, &t3); /* xi c2^2 - c0 c1 */
    fq2_sub(&s2, &t1, &t4); /* c1^2 - c0 c2 */
    fq2_mul(&a1, &x->c2, &s1);
    fq2_mul(&a3, &x->c1, &s2);
    fq2_add(&a1, &a1, &a3);
    fq2_mul_xi(&a1, &a1);
    fq2_mul(&a3, &x->c0, &s0);
    fq2_add(&a1, &a1, &a3);
    fq2_t inv;
    fq2_inv(&inv, &a1);
    fq2_mul(&r->c0, &s0, &inv);
    fq2_mul(&r->c1, &s1, &inv);
    fq2_mul(&r->c2, &s2, &inv);
}
static void fq6_frob(fq6_t* r, const fq6_t* x, int k)
{
    fq2_t c0 = x->c0, c1 = x->c1, c2 = x->c2;
    if (k & 1) {
        fq2_conj(&c0, &c0);
        fq2_conj(&c1, &c1);
        fq2_conj(&c2, &c2);
    }
    r->c0 = c0;
    fq2_mul(&r->c1, &c1, &PR_FROB6_C1[k]);
    fq2_mul(&r->c2, &c2, &PR_FROB6_C2[k]);
}

/* ---- Fq12 = Fq6[w]/(w^2 - v)  (ark-ff fp12_2over3over2.rs) */
static void fq12_one(fq12_t* r)
{
    memset(r, 0, sizeof *r);
    fq2_one(&r->c0.c0);
}
static int fq12_eq(const fq12_t* x, const fq12_t* y) { return memcmp(x, y, sizeof *x) == 0; } /* canonical limbs */
static void fq12_mul(fq12_t* r, const fq12_t* x, const fq12_t* y)
{
    fq6_t v0, v1, t, s0, s1;
    fq6_mul(&v0, &x->c0, &y->c0);
    fq6_mul(&v1, &x->c1, &y->c1);
    fq6_add(&s0, &x->c0, &x->c1);
    fq6_add(&s1, &y->c0, &y->c1);
    fq6_mul(&t, &s0, &s1);
    fq6_sub(&t, &t, &v0);
    fq6_sub(&t, &t, &v1);
    fq6_mul_v(&s0, &v1);
    fq6_add(&r->c0, &v0, &s0);
    r->c1 = t;
}
static void fq12_sqr(fq12_t* r, const fq12_t* x) { fq12_mul(r, x, x); }
static void fq12_conj(fq12_t* r, const fq12_t* x) /* x^(p^6): "cyclotomic inverse" for unitary elements */
{
    r->c0 = x->c0;
    fq6_neg(&r->c1, &x->c1);
}
static void fq12_inv(fq12_t* r, const fq12_t* x)
{
    /* 1/(c0 + c1 w) = (c0 - c1 w)/(c0^2 - v c1^2) */
    fq6_t t0, t1;
    fq6_mul(&t0, &x->c0, &x->c0);
    fq6_mul(&t1, &x->c1, &x->c1);
    fq6_mul_v(&t1, &t1);
    fq6_sub(&t0, &t0, &t1);
    fq6_inv(&t1, &t0);
    fq6_mul(&r->c0, &x->c0, &t1);
    fq6_mul(&t0, &x->c1, &t1);
    fq6_neg(&r->c1, &t0);
}
static void fq12_frob(fq12_t* r, const fq12_t* x, int k)
{
    fq6_t c0, c1;
    fq6_frob(&c0, &x->c0, k);
    fq6_frob(&c1, &x->c1, k);
    fq2_mul(&c1.c0, &c1.c0, &PR_FROB12_C1[k]);
    fq2_mul(&c1.c1, &c1.c1, &PR_FROB12_C1[k]);
    fq2_mul(&c1.c2, &c1.c2, &PR_FROB12_C1[k]);
    r->c0 = c0;
    r->c1 = c1;
}
/* f^x, x = 4965661367192848881 (ark-bn254 Config::X), then inverted: exp_by_neg_x (X_IS_NEGATIVE = false) */
static void fq12_exp_by_neg_x(fq12_t* r, const fq12_t* f)
{
    const u64 X = 4965661367192848881ull;
    fq12_t    acc;
    fq12_one(&acc);
    for (int i = 63; i >= 0; i--) {
        fq12_sqr(&acc, &acc);
        if ((X >> i) & 1) fq12_mul(&acc, &acc, f);
    }
    fq12_conj(r, &acc);
}

/* ---- line functions (ark-ec models/bn/g2.rs), D-type twist: coefficients (c0, c1, c2) = (ell_0, ell_vw, ell_vv) */
typedef struct { fq2_t x, y, z; } g2hom_t;
typedef struct { fq2_t c0, c1, c2; } ell_t;

static void g2hom_double(g2hom_t* r, ell_t* l)
{
    fq2_t a, b, c, e, f, g, h, i, j, e2, t;
    fq2_mul(&a, &r->x, &r->y);
    fq2_mul_fe(&a, &a, &PR_TWO_INV);
    fq2_sqr(&b, &r->y);
    fq2_sqr(&c, &r->z);
    fq2_dbl(&t, &c);
    fq2_add(&t, &t, &c);
    fq2_mul(&e, &PR_TWIST_B, &t);
    fq2_dbl(&f, &e);
    fq2_add(&f, &f, &e);
    fq2_add(&g, &b, &f);
    fq2_mul_fe(&g, &g, &PR_TWO_INV);
    fq2_add(&t, &r->y, &r->z);
    fq2_sqr(&h, &t);
    fq2_add(&t, &b, &c);
    fq2_sub(&h, &h, &t);
    fq2_sub(&i, &e, &b);
    fq2_sqr(&j, &r->x);
    fq2_sqr(&e2, &e);
    fq2_sub(&t, &b, &f);
    fq2_mul(&r->x, &a, &t);
    fq2_sqr(&t, &g);
    fq2_t e3;
    fq2_dbl(&e3, &e2);
    fq2_add(&e3, &e3, &e2);
    fq2_sub(&r->y, &t, &e3);
    fq2_mul(&r->z, &b, &h);
    fq2_neg(&l->c0, &h);
    fq2_dbl(&t, &j);
    fq2_add(&l->c1, &t, &j);
    l->c2 = i;
}
static void g2hom_add(g2hom_t* r, const g2_aff_t* q, ell_t* l)
{
    fq2_t theta, lambda, c, d, e, f, g, h, t, j;
    fq2_mul(&t, &q->y, &r->z);
    fq2_sub(&theta, &r->y, &t);
    fq2_mul(&t, &q->x, &r->z);
    fq2_sub(&lambda, &r->x, &t);
    fq2_sqr(&c, &theta);
    fq2_sqr(&d, &lambda);
    fq2_mul(&e, &lambda, &d);
    fq2_mul(&f, &r->z, &c);
    fq2_mul(&g, &r->x, &d);
    fq2_add(&h, &e, &f);
    fq2_dbl(&t, &g);
    fq2_sub(&h, &h, &t);
    fq2_mul(&r->x, &lambda, &h);
    fq2_sub(&t, &g, &h);
    fq2_mul(&t, &theta, &t);
    fq2_t ey;
    fq2_mul(&ey, &e, &r->y);
    fq2_sub(&r->y, &t, &ey);
    fq2_mul(&r->z, &r->z, &e);
    fq2_mul(&j, &theta, &q->x);
    fq2_mul(&t, &lambda, &q->y);
    fq2_sub(&j, &j, &t);
    l->c0 = lambda;
    fq2_neg(&l->c1, &theta);
    l->c2 = j;
}
/* models/bn/mod.rs ell(), D twist: f *= (c0 * p.y) + (c1 * p.x) v w + c2 v^2 ... as the sparse element
 * (c0*p.y, 0, 0) + (c1*p.x, c2, 0) w   ("mul_by_034") -- written here as a full multiplication by that element */
static void fq12_ell(fq12_t* f, const ell_t* l, const g1_aff_t* p)
{
    fq12_t s;
    memset(&s, 0, sizeof s);
    fq2_mul_fe(&s.c0.c0, &l->c0, &p->y);
    fq2_mul_fe(&s.c1.c0, &l->c1, &p->x);
    s.c1.c1 = l->c2;
    fq12_mul(f, f, &s);
}
static void g2_mul_by_char(g2_aff_t* r, const g2_aff_t* q)
{
    fq2_t x, y;
    fq2_conj(&x, &q->x);
    fq2_conj(&y, &q->y);
    fq2_mul(&r->x, &x, &PR_TWQX);
    fq2_mul(&r->y, &y, &PR_TWQY);
}
static const int8_t PR_ATE_NAF[65] = {0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0,
                                      0, 1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0,
                                      -1, 0, 0, 1, 0, 1, 1}; /* ark-bn254 Config::ATE_LOOP_COUNT = signed digits of 6x + 2 */

/* multi_miller_loop for ONE pair (pairs are independent factors); a zero P or Q contributes 1 */
static void pr_miller(fq12_t* f, const g1_aff_t* p, const g2_aff_t* q)
{
    pr_init();
    fq12_one(f);
    if (g1_aff_is_zero(p) || g2_aff_is_zero(q)) return;
    g2hom_t  r;
    g2_aff_t nq = *q;
    fq2_neg(&nq.y, &q->y);
    r.x = q->x;
    r.y = q->y;
    fq2_one(&r.z);
    ell_t l;
    for (int i = 64; i >= 1; i--) {
        if (i != 64) fq12_sqr(f, f);
        g2hom_double(&r, &l);
        fq12_ell(f, &l, p);
        int bit = PR_ATE_NAF[i - 1];
        if (bit == 1) {
            g2hom_add(&r, q, &l);
            fq12_ell(f, &l, p);
        } else if (bit == -1) {
            g2hom_add(&r, &nq, &l);
            fq12_ell(f, &l, p);
        }
    }
    g2_aff_t q1, q2;
    g2_mul_by_char(&q1, q);
    g2_mul_by_char(&q2, &q1);
    fq2_neg(&q2.y, &q2.y);
    g2hom_add(&r, &q1, &l);
    fq12_ell(f, &l, p);
    g2hom_add(&r, &q2, &l);
    fq12_ell(f, &l, p);
}
/* models/bn/mod.rs final_exponentiation */
static int pr_final_exp(fq12_t* out, const fq12_t* f)
{
    pr_init();
    fq12_t f1, f2, r, y0, y1, y2, y3, y4, y5, y6, y7, y8, y9, y10, y11, y12, y13, y14, y15;
    fq12_conj(&f1, f);
    {
        /* f = 0 has no inverse (arkworks returns None); cannot happen for Miller-loop outputs */
        fq12_t z;
        memset(&z, 0, sizeof z);
        if (fq12_eq(f, &z)) return -1;
    }
    fq12_inv(&f2, f);
    fq12_mul(&r, &f1, &f2);
    f2 = r;
    fq12_frob(&r, &r, 2);
    fq12_mul(&r, &r, &f2);
    fq12_exp_by_neg_x(&y0, &r);
    fq12_sqr(&y1, &y0);
    fq12_sqr(&y2, &y1);
    fq12_mul(&y3, &y2, &y1);
    fq12_exp_by_neg_x(&y4, &y3);
    fq12_sqr(&y5, &y4);
    fq12_exp_by_neg_x(&y6, &y5);
    fq12_conj(&y3, &y3);
    fq12_conj(&y6, &y6);
    fq12_mul(&y7, &y6, &y4);
    fq12_mul(&y8, &y7, &y3);
    fq12_mul(&y9, &y8, &y1);
    fq12_mul(&y10, &y8, &y4);
    fq12_mul(&y11, &y10, &r);
    fq12_frob(&y12, &y9, 1);
    fq12_mul(&y13, &y12, &y11);
    fq12_frob(&y8, &y8, 2);
    fq12_mul(&y14, &y8, &y13);
    fq12_conj(&r, &r);
    fq12_mul(&y15, &r, &y9);
    fq12_frob(&y15, &y15, 3);
    fq12_mul(out, &y15, &y14);
    return 0;
}

#endif
