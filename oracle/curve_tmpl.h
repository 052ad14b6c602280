/*
 * oracle/curve_tmpl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Short-Weierstrass y^2 = x^3 + b (a = 0) arithmetic in XYZZ coordinates,
 * written once and instantiated for G1 (over Fq) and G2 (over Fq2) by
 * including this file with the macros below defined:
 *
 *   CV(name)   symbol prefixer (g1_##name / g2_##name)
 *   FE         base-field element type
 *   F_MUL(r,a,b) F_SQR(r,a) F_ADD(r,a,b) F_SUB(r,a,b) F_NEG(r,a)
 *   F_ISZERO(a) F_EQ(a,b) F_ONE(r) F_ZERO(r) F_INV(r,a)
 *
 * Restates (paths relative to /root/reference/rust-rapidsnark/rapidsnark/src/):
 *   Point / PointAffine layout, infinity conventions ..... curve.hpp:18-30, curve.cpp:39-44, 532-539
 *   add(Point,Point,Point)        (EFD add-2008-s) ....... curve.cpp:91-166
 *   add(Point,Point,PointAffine)  (EFD madd-2008-s) ...... curve.cpp:185-250
 *   add(Point,Affine,Affine) ............................. curve.cpp:266-322
 *   dbl(Point,Point), dbl(Point,Affine) .................. curve.cpp:340-396, 411-458
 *   copy / neg / to-affine (field division) .............. curve.cpp:541-620
 *   nafMulByScalar / buildNaf ............................ exp.hpp:9-31, naf.cpp:55-74
 *   ParallelMultiexp (Pippenger, unsigned windows) ....... multiexp.cpp:26-71, 109-245
 */

typedef struct { FE x, y; } CV(aff_t);
typedef struct { FE x, y, zz, zzz; } CV(pt_t);

static inline int CV(aff_is_zero)(const CV(aff_t)* p) { return F_ISZERO(&p->x) && F_ISZERO(&p->y); }
static inline int CV(is_zero)(const CV(pt_t)* p) { return F_ISZERO(&p->zz); }

static inline void CV(set_zero)(CV(pt_t)* r)
{
    F_ONE(&r->x);
    F_ONE(&r->y);
    F_ZERO(&r->zz);
    F_ZERO(&r->zzz);
}
/* curve.cpp:548-561 */
static inline void CV(from_aff)(CV(pt_t)* r, const CV(aff_t)* a)
{
    if (CV(aff_is_zero)(a)) {
        CV(set_zero)(r);
        return;
    }
    r->x = a->x;
    r->y = a->y;
    F_ONE(&r->zz);
    F_ONE(&r->zzz);
}

/* curve.cpp:340-396 (a = 0 branch) */
static void CV(dbl)(CV(pt_t)* p3, const CV(pt_t)* p1)
{
    if (CV(is_zero)(p1)) {
        *p3 = *p1;
        return;
    }
    FE U, V, W, S, M, tmp, X3, Y3;
    F_ADD(&U, &p1->y, &p1->y);
    F_SQR(&V, &U);
    F_MUL(&W, &U, &V);
    F_MUL(&S, &p1->x, &V);
    F_SQR(&M, &p1->x);
    F_ADD(&tmp, &M, &M);
    F_ADD(&M, &M, &tmp);
    F_SQR(&X3, &M);
    F_SUB(&X3, &X3, &S);
    F_SUB(&X3, &X3, &S);
    F_MUL(&tmp, &W, &p1->y);
    F_SUB(&Y3, &S, &X3);
    F_MUL(&Y3, &M, &Y3);
    F_SUB(&Y3, &Y3, &tmp);
    FE ZZ3, ZZZ3;
    F_MUL(&ZZ3, &V, &p1->zz);
    F_MUL(&ZZZ3, &W, &p1->zzz);
    p3->x   = X3;
    p3->y   = Y3;
    p3->zz  = ZZ3;
    p3->zzz = ZZZ3;
}

/* curve.cpp:411-458 (a = 0) */
static void CV(dbl_aff)(CV(pt_t)* p3, const CV(aff_t)* p1)
{
    if (CV(aff_is_zero)(p1)) {
        CV(set_zero)(p3);
        return;
    }
    FE U, V, W, S, M, tmp, X3, Y3;
    F_ADD(&U, &p1->y, &p1->y);
    F_SQR(&V, &U);
    F_MUL(&W, &U, &V);
    F_MUL(&S, &p1->x, &V);
    F_SQR(&M, &p1->x);
    F_ADD(&tmp, &M, &M);
    F_ADD(&M, &tmp, &M);
    F_SQR(&X3, &M);
    F_SUB(&X3, &X3, &S);
    F_SUB(&X3, &X3, &S);
    F_MUL(&tmp, &W, &p1->y);
    F_SUB(&Y3, &S, &X3);
    F_MUL(&Y3, &M, &Y3);
    F_SUB(&Y3, &Y3, &tmp);
    p3->x   = X3;
    p3->y   = Y3;
    p3->zz  = V;
    p3->zzz = W;
}

/* curve.cpp:91-166 */
static void CV(add)(CV(pt_t)* p3, const CV(pt_t)* p1, const CV(pt_t)* p2)
{
    if (CV(is_zero)(p1)) {
        *p3 = *p2;
        return;
    }
    if (CV(is_zero)(p2)) {
        *p3 = *p1;
        return;
    }
    FE U1, U2, S1, S2, P, R, PP, PPP, Q, tmp, X3, Y3, ZZ3, ZZZ3;
    F_MUL(&U1, &p1->x, &p2->zz);
    F_MUL(&U2, &p2->x, &p1->zz);
    F_MUL(&S1, &p1->y, &p2->zzz);
    F_MUL(&S2, &p2->y, &p1->zzz);
    F_SUB(&P, &U2, &U1);
    F_SUB(&R, &S2, &S1);
    if (F_ISZERO(&P) && F_ISZERO(&R)) {
        CV(dbl)(p3, p1);
        return;
    }
    F_SQR(&PP, &P);
    F_MUL(&PPP, &P, &PP);
    F_MUL(&Q, &U1, &PP);
    F_SQR(&X3, &R);
    F_SUB(&X3, &X3, &PPP);
    F_SUB(&X3, &X3, &Q);
    F_SUB(&X3, &X3, &Q);
    F_MUL(&tmp, &S1, &PPP);
    F_SUB(&Y3, &Q, &X3);
    F_MUL(&Y3, &Y3, &R);
    F_SUB(&Y3, &Y3, &tmp);
    F_MUL(&ZZ3, &p1->zz, &p2->zz);
    F_MUL(&ZZ3, &ZZ3, &PP);
    F_MUL(&ZZZ3, &p1->zzz, &p2->zzz);
    F_MUL(&ZZZ3, &ZZZ3, &PPP);
    p3->x   = X3;
    p3->y   = Y3;
    p3->zz  = ZZ3;
    p3->zzz = ZZZ3;
}

/* curve.cpp:185-250 */
static void CV(madd)(CV(pt_t)* p3, const CV(pt_t)* p1, const CV(aff_t)* p2)
{
    if (CV(is_zero)(p1)) {
        CV(from_aff)(p3, p2);
        return;
    }
    if (CV(aff_is_zero)(p2)) {
        *p3 = *p1;
        return;
    }
    FE U2, S2, P, R, PP, PPP, Q, tmp, X3, Y3, ZZ3, ZZZ3;
    F_MUL(&U2, &p2->x, &p1->zz);
    F_MUL(&S2, &p2->y, &p1->zzz);
    F_SUB(&P, &U2, &p1->x);
    F_SUB(&R, &S2, &p1->y);
    if (F_ISZERO(&P) && F_ISZERO(&R)) {
        CV(dbl_aff)(p3, p2);
        return;
    }
    F_SQR(&PP, &P);
    F_MUL(&PPP, &P, &PP);
    F_MUL(&Q, &p1->x, &PP);
    F_SQR(&X3, &R);
    F_SUB(&X3, &X3, &PPP);
    F_SUB(&X3, &X3, &Q);
    F_SUB(&X3, &X3, &Q);
    F_MUL(&tmp, &p1->y, &PPP);
    F_SUB(&Y3, &Q, &X3);
    F_MUL(&Y3, &Y3, &R);
    F_SUB(&Y3, &Y3, &tmp);
    F_MUL(&ZZ3, &p1->zz, &PP);
    F_MUL(&ZZZ3, &p1->zzz, &PPP);
    p3->x   = X3;
    p3->y   = Y3;
    p3->zz  = ZZ3;
    p3->zzz = ZZZ3;
}

/* curve.cpp:266-322 */
static void CV(add_aff_aff)(CV(pt_t)* p3, const CV(aff_t)* p1, const CV(aff_t)* p2)
{
    if (CV(aff_is_zero)(p1)) {
        CV(from_aff)(p3, p2);
        return;
    }
    if (CV(aff_is_zero)(p2)) {
        CV(from_aff)(p3, p1);
        return;
    }
    FE P, R, PP, PPP, Q, tmp, X3, Y3;
    F_SUB(&P, &p2->x, &p1->x);
    F_SUB(&R, &p2->y, &p1->y);
    if (F_ISZERO(&P) && F_ISZERO(&R)) {
        CV(dbl_aff)(p3, p2);
        return;
    }
    F_SQR(&PP, &P);
    F_MUL(&PPP, &P, &PP);
    F_MUL(&Q, &p1->x, &PP);
    F_SQR(&X3, &R);
    F_SUB(&X3, &X3, &PPP);
    F_SUB(&X3, &X3, &Q);
    F_SUB(&X3, &X3, &Q);
    F_MUL(&tmp, &p1->y, &PPP);
    F_SUB(&Y3, &Q, &X3);
    F_MUL(&Y3, &Y3, &R);
    F_SUB(&Y3, &Y3, &tmp);
    p3->x   = X3;
    p3->y   = Y3;
    p3->zz  = PP;
    p3->zzz = PPP;
}

static inline void CV(neg)(CV(pt_t)* r, const CV(pt_t)* a)
{
    FE y;
    F_NEG(&y, &a->y);
    *r   = *a;
    r->y = y;
}
static inline void CV(neg_aff)(CV(aff_t)* r, const CV(aff_t)* a)
{
    FE y;
    F_NEG(&y, &a->y);
    r->x = a->x;
    r->y = y;
}
/* curve.hpp:121-133 : sub = add(neg) */
static inline void CV(sub)(CV(pt_t)* p3, const CV(pt_t)* p1, const CV(pt_t)* p2)
{
    CV(pt_t) t;
    CV(neg)(&t, p2);
    CV(add)(p3, p1, &t);
}
static inline void CV(msub)(CV(pt_t)* p3, const CV(pt_t)* p1, const CV(aff_t)* p2)
{
    CV(aff_t) t;
    CV(neg_aff)(&t, p2);
    CV(madd)(p3, p1, &t);
}

/* curve.cpp:565-576 : x = X/ZZ, y = Y/ZZZ ; infinity -> (0,0) */
static void CV(to_aff)(CV(aff_t)* r, const CV(pt_t)* a)
{
    if (CV(is_zero)(a)) {
        F_ZERO(&r->x);
        F_ZERO(&r->y);
        return;
    }
    FE i;
    F_INV(&i, &a->zz);
    F_MUL(&r->x, &a->x, &i);
    F_INV(&i, &a->zzz);
    F_MUL(&r->y, &a->y, &i);
}

/* curve.cpp:460-492 */
static int CV(eq)(const CV(pt_t)* p1, const CV(pt_t)* p2)
{
    if (CV(is_zero)(p1)) return CV(is_zero)(p2);
    if (CV(is_zero)(p2)) return 0;
    FE U1, U2, S1, S2;
    F_MUL(&U1, &p1->x, &p2->zz);
    F_MUL(&U2, &p2->x, &p1->zz);
    F_MUL(&S1, &p1->y, &p2->zzz);
    F_MUL(&S2, &p2->y, &p1->zzz);
    return F_EQ(&U1, &U2) && F_EQ(&S1, &S2);
}

/*
 * Single scalar multiplication, exp.hpp:9-31 + naf.cpp:55-74.
 * The reference builds a non-adjacent form with a byte-table; the digits are
 * the standard NAF of the scalar (digit in {0, 1, -1}), consumed MSB first as
 * dbl / add / sub.  The resulting group element is identical for any correct
 * recoding; this restatement computes the NAF directly.
 */
static void CV(mul_scalar_aff)(CV(pt_t)* r, const CV(aff_t)* base, const uint8_t* scalar, unsigned scalar_size)
{
    int      nbits = (int)scalar_size * 8 + 2;
    int8_t   naf[8 * 34 + 8];
    uint64_t k[5] = {0, 0, 0, 0, 0};
    memcpy(k, scalar, scalar_size > 32 ? 32 : scalar_size);
    for (int i = 0; i < nbits; i++) {
        int8_t d = 0;
        if (k[0] & 1) {
            d = (k[0] & 2) ? -1 : 1;
            if (d == 1) {
                k[0] &= ~(u64)1;
            } else { /* k += 1 */
                for (int j = 0; j < 5; j++) {
                    if (++k[j] != 0) break;
                }
            }
        }
        naf[i] = d;
        for (int j = 0; j < 4; j++) k[j] = (k[j] >> 1) | (k[j + 1] << 63);
        k[4] >>= 1;
    }
    CV(aff_t) b = *base;
    CV(set_zero)(r);
    int i = nbits - 1;
    while (i >= 0 && naf[i] == 0) i--;
    for (; i >= 0; i--) {
        CV(dbl)(r, r);
        if (naf[i] == 1)
            CV(madd)(r, r, &b);
        else if (naf[i] == -1)
            CV(msub)(r, r, &b);
    }
}
/* same for an XYZZ base (groth16.cpp:340-346 multiplies the XYZZ points pi_a, pib1) */
static void CV(mul_scalar)(CV(pt_t)* r, const CV(pt_t)* base, const uint8_t* scalar, unsigned scalar_size)
{
    /* double-and-add over XYZZ; group result equals the NAF walk */
    CV(pt_t) b = *base, acc;
    CV(set_zero)(&acc);
    for (int i = (int)scalar_size * 8 - 1; i >= 0; i--) {
        CV(dbl)(&acc, &acc);
        if (scalar[i >> 3] & (1u << (i & 7))) CV(add)(&acc, &acc, &b);
    }
    *r = acc;
}

/* multiexp.cpp:26-41 */
static inline u64 CV(get_chunk)(const uint8_t* scalars, u64 scalar_size, u64 bits_per_chunk, u64 scalar_idx,
                                u64 chunk_idx)
{
    u64 bit_start  = chunk_idx * bits_per_chunk;
    u64 byte_start = bit_start / 8;
    u64 eff        = bits_per_chunk;
    if (byte_start > scalar_size - 8) byte_start = scalar_size - 8;
    if (bit_start + bits_per_chunk > scalar_size * 8) eff = scalar_size * 8 - bit_start;
    u64 shift = bit_start - byte_start * 8;
    u64 v;
    memcpy(&v, scalars + scalar_idx * scalar_size + byte_start, 8);
    v >>= shift;
    v &= ((u64)1 << eff) - 1;
    return v;
}

/* multiexp.cpp:133-180 : sum_d d*bucket[d] by recursive halving (single thread) */
static void CV(bucket_reduce)(CV(pt_t)* res, CV(pt_t)* accs, u64 nbits)
{
    if (nbits == 1) {
        *res = accs[1];
        CV(set_zero)(&accs[1]);
        return;
    }
    u64      ndiv2 = (u64)1 << (nbits - 1);
    CV(pt_t) sall;
    memset(&sall, 0, sizeof sall); /* zz == 0 => infinity (multiexp.cpp:146) */
    for (u64 i = 0; i < ndiv2; i++) {
        if (!CV(is_zero)(&accs[ndiv2 + i])) {
            CV(add)(&accs[i], &accs[i], &accs[ndiv2 + i]);
            CV(add)(&sall, &sall, &accs[ndiv2 + i]);
            CV(set_zero)(&accs[ndiv2 + i]);
        }
    }
    CV(add)(&accs[ndiv2], &accs[ndiv2], &sall);
    CV(pt_t) p1;
    CV(bucket_reduce)(&p1, accs, nbits - 1);
    for (u64 i = 0; i < nbits - 1; i++) CV(dbl)(&accs[ndiv2], &accs[ndiv2]);
    CV(add)(res, &p1, &accs[ndiv2]);
    CV(set_zero)(&accs[ndiv2]);
}

/* multiexp.cpp:206-213 */
static inline unsigned CV(window_bits)(u64 n)
{
    uint32_t v = (uint32_t)(n / 2);
    unsigned c = 0;
    while (v > 1) {
        v >>= 1;
        c++;
    }
    if (c > 16) c = 16;
    if (c < 2) c = 2;
    return c;
}

/*
 * multiexp.cpp:183-245.  The reference parallelises processChunk over POINTS with one bucket array per thread
 * (multiexp.cpp:46-71, indexed by current_thread_index()) and then packs the thread copies (packThreads, :109-130).
 * Here the (optional) OpenMP parallelism is over (window, slice of the points) pairs, each with a private bucket
 * array, followed by the same pack step and the reference's reduction per window -- so the port scales to any core
 * count like the TBB loop does (nthreads = 1 is the plain serial algorithm).
 */
static void CV(msm)(CV(pt_t)* r, const CV(aff_t)* bases, const uint8_t* scalars, u64 scalar_size, u64 n,
                    int nthreads)
{
    if (n == 0) {
        CV(set_zero)(r);
        return;
    }
    if (n == 1) {
        CV(mul_scalar_aff)(r, &bases[0], scalars, (unsigned)scalar_size);
        return;
    }
    u64 c            = CV(window_bits)(n);
    u64 nchunks      = ((scalar_size * 8 - 1) / c) + 1;
    u64 accs_per     = (u64)1 << c;
    if (nthreads < 1) nthreads = 1;
    /* slices per window: enough (window, slice) tasks for every thread, but every slice keeps at least 4 points per
     * bucket -- each extra slice costs 2^c bucket initialisations and 2^c pack additions per window, so beyond that the
     * pack step outweighs the points it spreads (measured on 256 cores: 16 slices 0.5 M points/s, 4 slices 3x that) */
    u64 nslices = ((u64)nthreads + nchunks - 1) / nchunks;
    if (nslices > n / (4 * accs_per)) nslices = n / (4 * accs_per);
    if (nslices < 1) nslices = 1;
    CV(pt_t)* chunks = (CV(pt_t)*)malloc(nchunks * sizeof(CV(pt_t)));
    CV(pt_t)* accs   = (CV(pt_t)*)malloc(nchunks * nslices * accs_per * sizeof(CV(pt_t)));
    const int64_t ntasks = (int64_t)(nchunks * nslices);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t t = 0; t < ntasks; t++) {
        const u64 w = (u64)t / nslices, s = (u64)t % nslices;
        CV(pt_t)* a = accs + (u64)t * accs_per;
        for (u64 i = 0; i < accs_per; i++) CV(set_zero)(&a[i]);
        const u64 lo = n * s / nslices, hi = n * (s + 1) / nslices;
        for (u64 i = lo; i < hi; i++) {
            if (CV(aff_is_zero)(&bases[i])) continue;
            u64 d = CV(get_chunk)(scalars, scalar_size, c, i, w);
            if (d) CV(madd)(&a[d], &a[d], &bases[i]);
        }
    }
    /* packThreads (multiexp.cpp:109-130): bucket d of slice 0 += bucket d of every other slice; like the reference's
     * parallel_for over buckets, in blocks of 1024 buckets per task */
    if (nslices > 1) {
        const int64_t nblk = (int64_t)((accs_per + 1023) / 1024);
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(nthreads)
        for (int64_t w = 0; w < (int64_t)nchunks; w++)
            for (int64_t blk = 0; blk < nblk; blk++) {
                CV(pt_t)* a0 = accs + (u64)w * nslices * accs_per;
                u64 d0 = (u64)blk * 1024, d1 = d0 + 1024 < accs_per ? d0 + 1024 : accs_per;
                for (u64 s = 1; s < nslices; s++) {
                    CV(pt_t)* as = a0 + s * accs_per;
                    for (u64 d = d0 ? d0 : 1; d < d1; d++)
                        if (!CV(is_zero)(&as[d])) CV(add)(&a0[d], &a0[d], &as[d]);
                }
            }
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t w = 0; w < (int64_t)nchunks; w++) CV(bucket_reduce)(&chunks[w], accs + (u64)w * nslices * accs_per, c);
    free(accs);
    *r = chunks[nchunks - 1];
    for (int64_t j = (int64_t)nchunks - 2; j >= 0; j--) {
        for (u64 k = 0; k < c; k++) CV(dbl)(r, r);
        CV(add)(r, r, &chunks[j]);
    }
    free(chunks);
}
