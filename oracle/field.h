/*
 * oracle/field.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, 4x64-bit limbs, unsigned __int128) of the BN254
 * base field Fq and scalar field Fr used on the Groth16 hot path of the
 * reference (aptos-labs/keyless-zk-proofs, vendored rapidsnark).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product path (keyless-zk-proofs_amd/) never does.
 *
 * Reference behaviour restated here (paths relative to
 * /root/reference/rust-rapidsnark/rapidsnark/src/):
 *   constants q, R^2, -q^-1 mod 2^64 ........ fq_raw_generic.cpp:6-8, fr_raw_generic.cpp:5-7
 *   R^3 ..................................... fq_generic.cpp:8, fr_generic.cpp:8
 *   rawAdd / rawSub / rawNeg ................ fq_raw_generic.cpp:12-40, 69-81
 *   rawMMul (CIOS + one conditional subtract) fq_raw_generic.cpp:108-149
 *   to/fromMontgomery ....................... fq_raw_generic.cpp:193-233
 *   inv (= a^-1 in Montgomery form) ......... fq.cpp:238-250
 *   toString (fromMontgomery, base 10) ...... fq.cpp:225-236
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H

#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 v[4]; } fe_t; /* little-endian limbs; Montgomery form unless stated */

typedef struct {
    u64  p[4];
    u64  np; /* -p^-1 mod 2^64 */
    fe_t r2; /* R^2 mod p */
    fe_t r3; /* R^3 mod p */
    fe_t one; /* R mod p */
} fparams_t;

extern const fparams_t ORA_FQ;
extern const fparams_t ORA_FR;

static inline int fe_is_zero(const fe_t* a)
{
    return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0;
}
static inline int fe_eq(const fe_t* a, const fe_t* b)
{
    return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
/* a >= b on raw 256-bit integers */
static inline int raw_geq(const u64* a, const u64* b)
{
    for (int i = 3; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return 1;
}
static inline u64 raw_add(u64* r, const u64* a, const u64* b)
{
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
static inline u64 raw_sub(u64* r, const u64* a, const u64* b)
{
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i]   = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}

/* fq_raw_generic.cpp:12-20 : reduce on carry OR >= p */
static inline void fe_add(const fparams_t* F, fe_t* r, const fe_t* a, const fe_t* b)
{
    u64 t[4];
    u64 carry = raw_add(t, a->v, b->v);
    if (carry || raw_geq(t, F->p)) raw_sub(t, t, F->p);
    memcpy(r->v, t, 32);
}
/* fq_raw_generic.cpp:32-40 : add p back on borrow */
static inline void fe_sub(const fparams_t* F, fe_t* r, const fe_t* a, const fe_t* b)
{
    u64 t[4];
    u64 borrow = raw_sub(t, a->v, b->v);
    if (borrow) raw_add(t, t, F->p);
    memcpy(r->v, t, 32);
}
/* fq_raw_generic.cpp:69-81 : neg(0) = 0 */
static inline void fe_neg(const fparams_t* F, fe_t* r, const fe_t* a)
{
    if (fe_is_zero(a)) {
        memset(r, 0, sizeof *r);
    } else {
        u64 t[4];
        raw_sub(t, F->p, a->v);
        memcpy(r->v, t, 32);
    }
}

/* fq_raw_generic.cpp:108-149 : Montgomery product a*b*R^-1 mod p, R = 2^256.
 * Word-serial CIOS over 5-limb partial products, then ONE conditional
 * subtraction of the low four limbs.  For canonical inputs (< p, everything on
 * the proving path) no partial product overflows 5 limbs and the result is
 * in [0,p).  For operands >= p the reference's generic backend lets the carry
 * out of the 5th limb re-enter at limb 1 of the next partial product
 * (":121, :128, :135" store it in productK[1]) and drops the last one; that is
 * restated literally so the reference's own non-canonical KATs
 * (test_prover.cpp Fr_Rw_mul test 3) reproduce bit-for-bit. */
static inline void fe_mul(const fparams_t* F, fe_t* r, const fe_t* a, const fe_t* b)
{
    u64 t[5] = {0, 0, 0, 0, 0};
    u64 cprev = 0;
    for (int i = 0; i < 4; i++) {
        u64  nw[5] = {0, cprev, 0, 0, 0};
        u128 c     = 0;
        for (int j = 0; j < 4; j++) { /* nw[0..3] += b * a[i], carry -> nw[4] */
            c += (u128)a->v[i] * b->v[j] + nw[j];
            nw[j] = (u64)c;
            c >>= 64;
        }
        nw[4] = (u64)c;
        if (i > 0) { /* nw += t >> 64 (5-limb add, overflow dropped) */
            c = 0;
            for (int j = 0; j < 5; j++) {
                c += (u128)nw[j] + (j < 4 ? t[j + 1] : 0);
                nw[j] = (u64)c;
                c >>= 64;
            }
        }
        u64 m = nw[0] * F->np;
        c     = 0;
        for (int j = 0; j < 5; j++) { /* nw += p * m over 5 limbs */
            c += (u128)m * (j < 4 ? F->p[j] : 0) + nw[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        cprev = (u64)c;
    }
    u64 res[4] = {t[1], t[2], t[3], t[4]};
    if (raw_geq(res, F->p)) raw_sub(res, res, F->p);
    memcpy(r->v, res, 32);
}
static inline void fe_sqr(const fparams_t* F, fe_t* r, const fe_t* a) { fe_mul(F, r, a, a); }

/* fq_raw_generic.cpp:193-196 */
static inline void fe_to_mont(const fparams_t* F, fe_t* r, const fe_t* a) { fe_mul(F, r, a, &F->r2); }
/* fq_raw_generic.cpp:198-233 : MMul(a, 1) */
static inline void fe_from_mont(const fparams_t* F, fe_t* r, const fe_t* a)
{
    fe_t one = {{1, 0, 0, 0}};
    fe_mul(F, r, a, &one);
}

void fe_pow(const fparams_t* F, fe_t* r, const fe_t* base, const u64 e[4]);
void fe_inv(const fparams_t* F, fe_t* r, const fe_t* a);
/* decimal string of the standard-form value of Montgomery element a; returns length */
int  fe_to_dec(const fparams_t* F, char* out, const fe_t* a);
/* decimal string -> Montgomery element (value reduced mod p) */
void fe_from_dec(const fparams_t* F, fe_t* r, const char* s);

#endif
