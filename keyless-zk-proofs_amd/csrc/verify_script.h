// verify_script.h -- host side of the wave-cooperative ("latency") Groth16 verifier: ONE wavefront per proof.
//
// The service verifies every proof right after proving it (prover-service/src/request_handler/prover_handler.rs:329-336).
// The batched verifier (one lane per pairing) needs ~45 ms however small the batch is, because one lane walks ~32 000
// dependent field multiplications.  A pairing check has plenty of independent work at every point -- an Fp12 product is
// 54..72 independent Fq products -- so here the whole check of one proof is laid out ONCE, on the host, as a static
// program for a 64-lane wavefront:
//
//   1. the tower code of bn254_pairing_body.inc (the SAME source the batched kernels execute) is run with a RECORDING
//      field type (namespace k16t): every Fq operation appends a node to an expression graph -- multi-Miller loop of the
//      three pairs (shared squaring, as ark-ec's multi_miller_loop; the two pairs with fixed G2 arguments -gamma, -delta
//      read their line coefficients from a per-key table, as ark-groth16's prepared key does) and the final exponentiation;
//   2. dead nodes are dropped, chains of additions / subtractions / small multiples are flattened into linear combinations
//      of multiplication results, and the graph is list-scheduled into STEPS of at most 64 independent operations of one
//      class (multiply | linear combination | inversion), latest-start first so that values are produced just in time;
//   3. values get slots of an LDS register file by liveness.
//
// The device kernel (verify.hip, k_verify_coop) is an interpreter of that program: per step every lane fetches its
// instruction word from LDS, reads its operands from the slot file, executes, writes its result.  Field elements are
// canonical Montgomery values throughout, and field arithmetic is exact, so every value -- the GT element included --
// equals what the one-lane-per-pairing kernels and the oracle compute (tests/cpp/verify_script_check.cpp runs the program
// on the host against miller_loop / final_exponentiation; tests/test_gpu_verify.py compares the flags of both GPU paths).
#pragma once
#include <algorithm>
#include <map>
#include <stdexcept>
#include <stdint.h>
#include <unordered_map>
#include <vector>
#include "bn254_pairing.h"

namespace k16t {

// ------------------------------------------------------------------------------------------------ the recording field
enum : uint8_t { N_IN = 0, N_MUL, N_INV, N_ADD, N_SUB, N_NEG, N_DBL, N_XIA /* 9a - b */, N_XIB /* a + 9b */ };
struct Node {
    uint8_t op;
    int32_t a, b;
};
struct Builder {
    std::vector<Node>                     nodes;
    std::unordered_map<uint64_t, int32_t> cse;
    int32_t                               zero_id = -1, one_id = -1;
    int32_t add_node(uint8_t op, int32_t a, int32_t b)
    {
        if (op == N_MUL || op == N_ADD) // commutative
            if (a > b) std::swap(a, b);
        if (op != N_IN) {
            const uint64_t key = ((uint64_t)op << 58) ^ ((uint64_t)(uint32_t)a << 29) ^ (uint64_t)(uint32_t)b;
            auto           it  = cse.find(key);
            if (it != cse.end()) {
                const Node& n = nodes[it->second];
                if (n.op == op && n.a == a && n.b == b) return it->second;
            }
            nodes.push_back(Node{op, a, b});
            cse[key] = (int32_t)nodes.size() - 1;
            return (int32_t)nodes.size() - 1;
        }
        nodes.push_back(Node{op, a, b});
        return (int32_t)nodes.size() - 1;
    }
};
inline Builder*& cur()
{
    static thread_local Builder* b = nullptr;
    return b;
}

struct Fq {
    int32_t   id;
    static Fq zero() { return Fq{cur()->zero_id}; }
    static Fq one() { return Fq{cur()->one_id}; }
};
inline bool is0(const Fq& x) { return x.id == cur()->zero_id; }
inline bool is1(const Fq& x) { return x.id == cur()->one_id; }
inline Fq   fneg(const Fq& a) { return is0(a) ? a : Fq{cur()->add_node(N_NEG, a.id, -1)}; }
inline Fq   fdbl(const Fq& a) { return is0(a) ? a : Fq{cur()->add_node(N_DBL, a.id, -1)}; }
inline Fq   fadd(const Fq& a, const Fq& b)
{
    if (is0(a)) return b;
    if (is0(b)) return a;
    if (a.id == b.id) return fdbl(a);
    return Fq{cur()->add_node(N_ADD, a.id, b.id)};
}
inline Fq fsub(const Fq& a, const Fq& b)
{
    if (is0(b)) return a;
    if (a.id == b.id) return Fq::zero();
    if (is0(a)) return fneg(b);
    return Fq{cur()->add_node(N_SUB, a.id, b.id)};
}
inline Fq fmul(const Fq& a, const Fq& b)
{
    if (is0(a) || is0(b)) return Fq::zero();
    if (is1(a)) return b;
    if (is1(b)) return a;
    return Fq{cur()->add_node(N_MUL, a.id, b.id)};
}
inline Fq fsqr(const Fq& a) { return fmul(a, a); }
inline Fq finv(const Fq& a)
{
    if (is1(a)) return a;
    return Fq{cur()->add_node(N_INV, a.id, -1)};
}

// Fq2 = Fq[u]/(u^2 + 1).  Schoolbook products (4 multiplications, no operand sums): with 64 lanes the extra product is
// free and the multiplication level starts one linear level earlier than Karatsuba's.  Same values (exact arithmetic).
struct Fq2 {
    Fq         a, b;
    static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    bool       is_zero() const { return false; }                 // the generic case is recorded; degenerate inputs take the
    bool       operator==(const Fq2&) const { return false; }    // one-lane-per-pairing path (k16_verify_batch decides)
};
inline Fq2 fadd(const Fq2& x, const Fq2& y) { return Fq2{fadd(x.a, y.a), fadd(x.b, y.b)}; }
inline Fq2 fsub(const Fq2& x, const Fq2& y) { return Fq2{fsub(x.a, y.a), fsub(x.b, y.b)}; }
inline Fq2 fneg(const Fq2& x) { return Fq2{fneg(x.a), fneg(x.b)}; }
inline Fq2 fdbl(const Fq2& x) { return Fq2{fdbl(x.a), fdbl(x.b)}; }
inline Fq2 fmul(const Fq2& x, const Fq2& y)
{
    return Fq2{fsub(fmul(x.a, y.a), fmul(x.b, y.b)), fadd(fmul(x.a, y.b), fmul(x.b, y.a))};
}
inline Fq2 fsqr(const Fq2& x) { return Fq2{fsub(fsqr(x.a), fsqr(x.b)), fdbl(fmul(x.a, x.b))}; }
inline Fq2 finv(const Fq2& x)
{
    Fq t = finv(fadd(fsqr(x.a), fsqr(x.b)));
    return Fq2{fmul(x.a, t), fneg(fmul(x.b, t))};
}
inline Fq2 fmul_xi(const Fq2& x)
{
    auto xi = [](uint8_t op, const Fq& p, const Fq& q) -> Fq {
        if (is0(p) && is0(q)) return Fq::zero();
        return Fq{cur()->add_node(op, p.id, q.id)};
    };
    return Fq2{xi(N_XIA, x.a, x.b), xi(N_XIB, x.a, x.b)};
}
template <class F>
struct Aff {
    F    x, y;
    bool is_zero() const { return false; }
};

#pragma push_macro("K16_HD")
#pragma push_macro("K16_HDN")
#undef K16_HD
#undef K16_HDN
#define K16_HD inline
#define K16_HDN inline
#define K16_PAIRING_CUSTOM_XI
#include "bn254_pairing_body.inc"
#undef K16_PAIRING_CUSTOM_XI
#pragma pop_macro("K16_HDN")
#pragma pop_macro("K16_HD")

} // namespace k16t

namespace k16 {

// ------------------------------------------------------------------------------------------------ the program
// Instruction classes of a step
enum : uint8_t { CS_MUL = 1, CS_LIN = 2, CS_INV = 3 };
// One lane's instruction of a step, 64 bits:
//   MUL  dst[0:14) | a[14:28) | b[28:42) | valid[63]
//   INV  dst[0:14) | a[14:28)             | valid[63]
//   LIN  dst[0:14) | nterms[14:20) | first term[20:44) | valid[63]     terms: int16 coefficient << 16 | slot
//        (group k of a step sits in lanes 16 (k / 5) + 3 (k % 5) + {0, 1, 2}: lane + g handles limbs 3g .. 3g+2)
struct CoopProgram {
    // constant table layout (slots [0, n_const)): 0 = zero, 1 = one, then PairConsts, the target e(alpha, beta), and the
    // line coefficients of the two fixed pairs -- filled per key by coop_const_table()
    uint32_t n_const = 0, n_lines = 0;
    uint32_t in_base = 0;     // inputs: A.x A.y | B.x.a B.x.b B.y.a B.y.b | C.x C.y | vk_x as X ZZZ, Y ZZ, ZZ ZZZ  (COOP_N_INPUTS slots)
    uint32_t n_slots = 0;     // slot file size (constants + inputs + temporaries)
    uint32_t out_slot[12];    // the GT value e(A,B) e(vk_x,-gamma) e(C,-delta), c0.c0.a first
    uint32_t target_const = 0; // first of the 12 constant slots holding e(alpha, beta)
    std::vector<uint8_t>  step_class;
    std::vector<uint64_t> words;   // 64 per step
    std::vector<uint32_t> terms;
    // statistics
    uint32_t n_mul_ops = 0, n_lin_ops = 0, n_inv_ops = 0, depth = 0, mul_depth = 0;
};

constexpr uint32_t COOP_N_INPUTS = 11;
#ifndef COOP_TRIP_TERMS
#define COOP_TRIP_TERMS 8
#endif
constexpr uint32_t COOP_TRIP = COOP_TRIP_TERMS; // terms a lane accumulates per loop trip of a linear step (power of two)
// number of line evaluations of one Miller loop: a doubling per digit of 6x + 2 below the leading one, an addition per
// non-zero digit, two Frobenius additions
inline uint32_t coop_line_count()
{
    uint32_t n = 0;
    for (int i = (int)ATE_TOP; i >= 1; i--) {
        n++;
        if ((ATE_NZ_LO >> (unsigned)(i - 1)) & 1) n++;
    }
    return n + 2;
}

// The line coefficients of the Miller loop for a FIXED G2 point (ark-ec G2Prepared::from): lines[k] = (c0, c1, c2).
inline void coop_prepare_lines(const Aff<Fq2>& q, const PairConsts& K, std::vector<Ell>* out)
{
    out->clear();
    G2Hom    r{q.x, q.y, Fq2::one()};
    Aff<Fq2> nq{q.x, fneg(q.y)};
    Ell      l;
    for (int i = (int)ATE_TOP; i >= 1; i--) {
        g2hom_double(&r, &l, &K);
        out->push_back(l);
        const unsigned d = (unsigned)(i - 1);
        if ((ATE_NZ_LO >> d) & 1) {
            g2hom_add(&r, ((ATE_NEG_LO >> d) & 1) ? &nq : &q, &l);
            out->push_back(l);
        }
    }
    Aff<Fq2> q1 = g2_mul_by_char(q, K);
    Aff<Fq2> q2 = g2_mul_by_char(q1, K);
    q2.y        = fneg(q2.y);
    g2hom_add(&r, &q1, &l);
    out->push_back(l);
    g2hom_add(&r, &q2, &l);
    out->push_back(l);
}

// flattening order of PairConsts (the program's constant slots 2 ..): must match coop_const_table
constexpr uint32_t COOP_NPC = 2 * (1 + 2 + 4 + 4 + 4) + 1;
inline void coop_flatten_consts(const PairConsts& K, Fq* out /* [COOP_NPC] */)
{
    uint32_t n = 0;
    auto     p2 = [&](const Fq2& v) {
        out[n++] = v.a;
        out[n++] = v.b;
    };
    p2(K.twist_b);
    p2(K.twqx);
    p2(K.twqy);
    for (int k = 0; k < 4; k++) p2(K.frob6_c1[k]);
    for (int k = 0; k < 4; k++) p2(K.frob6_c2[k]);
    for (int k = 0; k < 4; k++) p2(K.frob12_c1[k]);
    out[n++] = K.two_inv;
}
// per key: [zero, one, PairConsts, e(alpha,beta) (12), lines of -gamma (6 each), lines of -delta]
inline void coop_const_table(const PairConsts& K, const Fp12& eab, const std::vector<Ell>& l1, const std::vector<Ell>& l2,
                             std::vector<Fq>* out)
{
    out->clear();
    out->push_back(Fq::zero());
    out->push_back(Fq::one());
    Fq pc[COOP_NPC];
    coop_flatten_consts(K, pc);
    for (uint32_t i = 0; i < COOP_NPC; i++) out->push_back(pc[i]);
    const Fq2* e = &eab.c0.c0;
    for (int i = 0; i < 6; i++) {
        out->push_back(e[i].a);
        out->push_back(e[i].b);
    }
    for (const std::vector<Ell>* L : {&l1, &l2})
        for (const Ell& l : *L) {
            const Fq2 c[3] = {l.c0, l.c1, l.c2};
            for (const Fq2& v : c) {
                out->push_back(v.a);
                out->push_back(v.b);
            }
        }
}

#ifndef COOP_WINDOW
#define COOP_WINDOW 8
#endif
#ifndef COOP_MAX_TERMS
#define COOP_MAX_TERMS 16 /* measured on MI355X, one proof: 12 -> 2.15 ms, 16 -> 2.07, 20 -> 2.32, 40 -> 2.57 (8 terms per trip) */
#endif
#ifndef COOP_MAX_COEF
#define COOP_MAX_COEF (1 << 12)
#endif
// what k_verify_coop's linear step is built around (verify.hip): a bias of 2^15 p under coefficients of at most COOP_MAX_COEF
// times values below 5p, int16 coefficients, a 6-bit term count, whole trips of a power-of-two number of terms
static_assert(COOP_MAX_COEF * 5 < (1 << 15), "linear step: the 2^15 p bias must cover the negative part of a combination");
static_assert(COOP_MAX_TERMS < 64, "linear step: 6-bit term count");
static_assert((COOP_TRIP_TERMS & (COOP_TRIP_TERMS - 1)) == 0, "linear step: terms per trip must be a power of two");
// Builds the program.  Depends only on the curve constants' zero / one pattern (K), not on a key.
inline void coop_build_program(const PairConsts& K, CoopProgram* P)
{
    namespace t = k16t;
    t::Builder B;
    t::cur() = &B;
    struct Reset {
        ~Reset() { t::cur() = nullptr; }
    } reset;
    const uint32_t n_lines = coop_line_count();
    // ---- constants and inputs are N_IN nodes; node id -> slot
    std::vector<int32_t> slot_of_in; // per N_IN node
    auto new_in = [&](int32_t slot) {
        int32_t id = B.add_node(t::N_IN, slot, -1);
        return t::Fq{id};
    };
    B.zero_id = new_in(0).id;
    B.one_id  = new_in(1).id;
    Fq pc[COOP_NPC];
    coop_flatten_consts(K, pc);
    uint32_t next = 2;
    auto     cst = [&](const Fq& concrete) -> t::Fq { // a constant slot; known zeros / ones fold
        const uint32_t s = next++;
        if (concrete.is_zero()) return t::Fq::zero();
        if (concrete == Fq::one()) return t::Fq::one();
        return new_in((int32_t)s);
    };
    t::PairConsts TK;
    {
        uint32_t n  = 0;
        auto     g2 = [&]() {
            t::Fq a = cst(pc[n]), b = cst(pc[n + 1]);
            n += 2;
            return t::Fq2{a, b};
        };
        TK.twist_b = g2();
        TK.twqx    = g2();
        TK.twqy    = g2();
        for (int k = 0; k < 4; k++) TK.frob6_c1[k] = g2();
        for (int k = 0; k < 4; k++) TK.frob6_c2[k] = g2();
        for (int k = 0; k < 4; k++) TK.frob12_c1[k] = g2();
        TK.two_inv = cst(pc[n]);
    }
    P->target_const = next;
    next += 12;
    const uint32_t line_base = next;
    next += 2 * n_lines * 6;
    P->n_const = next;
    P->n_lines = n_lines;
    P->in_base = next;
    auto inp = [&](uint32_t k) { return new_in((int32_t)(P->in_base + k)); };
    // vk_x comes in PROJECTIVE form (XYZZ: x = X / ZZ, y = Y / ZZZ), as sx = X ZZZ, sy = Y ZZ, sz = ZZ ZZZ: its lines are
    // evaluated scaled by sz, (c0 sy, c1 sx, c2 sz) instead of (c0 y, c1 x, c2).  The factor is in Fq, so the final
    // exponentiation removes it ((p - 1) divides (p^12 - 1) / r) -- and the prologue needs no inversion for vk_x.
    t::Aff<t::Fq>  pa{inp(0), inp(1)}, pc3{inp(6), inp(7)};
    const t::Fq    vk_sx = inp(8), vk_sy = inp(9), vk_sz = inp(10);
    t::Aff<t::Fq2> qb{t::Fq2{inp(2), inp(3)}, t::Fq2{inp(4), inp(5)}};
    next += COOP_N_INPUTS;
    auto line = [&](uint32_t pair /* 1 or 2 */, uint32_t k) {
        const uint32_t s = line_base + ((pair - 1) * n_lines + k) * 6;
        t::Ell         l;
        l.c0 = t::Fq2{new_in((int32_t)s), new_in((int32_t)s + 1)};
        l.c1 = t::Fq2{new_in((int32_t)s + 2), new_in((int32_t)s + 3)};
        l.c2 = t::Fq2{new_in((int32_t)s + 4), new_in((int32_t)s + 5)};
        return l;
    };
    // ---- the multi-Miller loop (ark-ec Bn::multi_miller_loop): one squaring per digit for all three pairs
    t::Fp12        f = t::f12_one();
    t::G2Hom       r{qb.x, qb.y, t::Fq2::one()};
    t::Aff<t::Fq2> nq{qb.x, t::fneg(qb.y)};
    t::Ell         l;
    uint32_t       k = 0;
    auto ells = [&]() { // the three line evaluations of one step; the fixed pairs first: their operands are ready early
        {
            const t::Ell l1 = line(1, k);
            t::Fq2       c0 = t::fmul_fp(l1.c0, vk_sy), c1 = t::fmul_fp(l1.c1, vk_sx), c2 = t::fmul_fp(l1.c2, vk_sz);
            t::f12_mul_by_034(&f, &c0, &c1, &c2);
        }
        t::f12_ell(&f, line(2, k), pc3);
        t::f12_ell(&f, l, pa);
        k++;
    };
    for (int i = (int)ATE_TOP; i >= 1; i--) {
        if (i != (int)ATE_TOP) t::f12_sqr(&f, &f);
        t::g2hom_double(&r, &l, &TK);
        ells();
        const unsigned d = (unsigned)(i - 1);
        if ((ATE_NZ_LO >> d) & 1) {
            t::g2hom_add(&r, ((ATE_NEG_LO >> d) & 1) ? &nq : &qb, &l);
            ells();
        }
    }
    t::Aff<t::Fq2> q1 = t::g2_mul_by_char(qb, TK);
    t::Aff<t::Fq2> q2 = t::g2_mul_by_char(q1, TK);
    q2.y              = t::fneg(q2.y);
    t::g2hom_add(&r, &q1, &l);
    ells();
    t::g2hom_add(&r, &q2, &l);
    ells();
    if (k != n_lines) throw std::logic_error("coop program: line count");
    t::Fp12 e;
    (void)t::final_exponentiation(&e, &f, &TK);
    const t::Fq2* ev = &e.c0.c0;
    int32_t       out_node[12];
    for (int i = 0; i < 6; i++) {
        out_node[2 * i]     = ev[i].a.id;
        out_node[2 * i + 1] = ev[i].b.id;
    }

    // ---- reachability
    const int32_t        N = (int32_t)B.nodes.size();
    std::vector<uint8_t> live(N, 0);
    {
        std::vector<int32_t> st(out_node, out_node + 12);
        while (!st.empty()) {
            int32_t v = st.back();
            st.pop_back();
            if (v < 0 || live[v]) continue;
            live[v]       = 1;
            const auto& n = B.nodes[v];
            if (n.op != t::N_IN) {
                st.push_back(n.a);
                if (n.b >= 0) st.push_back(n.b);
            }
        }
    }
    // ---- linear nodes as linear combinations of MATERIALISED nodes (inputs, products, inverses, and linear nodes that a
    // product / inverse / output consumes, or whose expansion would grow too long)
    typedef std::map<int32_t, int32_t> Comb; // node -> coefficient
    constexpr size_t  MAX_TERMS = COOP_MAX_TERMS;
    constexpr int32_t MAX_COEF  = COOP_MAX_COEF;
    // A linear node is kept as a combination of NON-LINEAR nodes (products, inverses, inputs) as long as that stays short --
    // then every value a product needs is ONE linear step after the products it is made of -- and otherwise of "atoms":
    // linear nodes that are materialised and referenced as a single term.
    std::vector<uint8_t> mat(N, 0), is_lin(N, 0), atom(N, 0);
    std::vector<Comb>    comb(N);
    for (int32_t v = 0; v < N; v++) {
        const uint8_t op = B.nodes[v].op;
        is_lin[v]        = op >= t::N_ADD;
        if (!is_lin[v]) mat[v] = atom[v] = 1;
    }
    for (int32_t v = 0; v < N; v++)
        if (live[v] && !is_lin[v] && B.nodes[v].op != t::N_IN) {
            mat[B.nodes[v].a] = 1;
            if (B.nodes[v].b >= 0) mat[B.nodes[v].b] = 1;
        }
    for (int i = 0; i < 12; i++) mat[out_node[i]] = 1;
    auto expansion = [&](int32_t v) -> Comb { // what v contributes when it is an OPERAND
        if (atom[v]) return Comb{{v, 1}};
        return comb[v];
    };
    auto coef_sum = [](const Comb& c) {
        int64_t s = 0;
        for (auto& kv : c) s += kv.second < 0 ? -(int64_t)kv.second : kv.second;
        return s;
    };
    for (int32_t v = 0; v < N; v++) {
        if (!live[v] || !is_lin[v]) continue;
        const auto& n = B.nodes[v];
        for (int attempt = 0;; attempt++) {
            Comb c;
            auto acc = [&](int32_t src, int32_t kf) {
                for (auto& kv : expansion(src)) {
                    c[kv.first] += kv.second * kf;
                    if (c[kv.first] == 0) c.erase(kv.first);
                }
            };
            switch (n.op) {
            case t::N_ADD: acc(n.a, 1); acc(n.b, 1); break;
            case t::N_SUB: acc(n.a, 1); acc(n.b, -1); break;
            case t::N_NEG: acc(n.a, -1); break;
            case t::N_DBL: acc(n.a, 2); break;
            case t::N_XIA: acc(n.a, 9); acc(n.b, -1); break;
            case t::N_XIB: acc(n.a, 1); acc(n.b, 9); break;
            default: throw std::logic_error("coop program: op");
            }
            if (c.size() <= MAX_TERMS && coef_sum(c) <= MAX_COEF) {
                comb[v] = c;
                break;
            }
            // too long: the operand with the longer expansion becomes an atom (a step of its own), and again
            int32_t pick = n.a;
            if (n.b >= 0 && !atom[n.b] && (atom[n.a] || comb[n.b].size() > comb[n.a].size())) pick = n.b;
            if (atom[pick] || attempt > 2) throw std::logic_error("coop program: cannot bound a linear combination");
            atom[pick] = mat[pick] = 1;
        }
    }
    // ---- operations: one per materialised live non-input node
    struct OpI {
        uint8_t              cls;
        int32_t              node;
        std::vector<int32_t> deps; // operand nodes (materialised)
        int32_t              level = 0, alap = 0, step = -1;
    };
    std::vector<OpI>     ops;
    std::vector<int32_t> op_of(N, -1);
    for (int32_t v = 0; v < N; v++) {
        if (!live[v] || !mat[v] || B.nodes[v].op == t::N_IN) continue;
        OpI o;
        o.node = v;
        if (is_lin[v]) {
            o.cls = CS_LIN;
            for (auto& kv : comb[v]) o.deps.push_back(kv.first);
            if (comb[v].empty()) throw std::logic_error("coop program: empty combination"); // (a value that is identically 0)
        } else {
            o.cls = B.nodes[v].op == t::N_MUL ? CS_MUL : CS_INV;
            o.deps.push_back(B.nodes[v].a);
            if (B.nodes[v].b >= 0) o.deps.push_back(B.nodes[v].b);
        }
        op_of[v] = (int32_t)ops.size();
        ops.push_back(o);
    }
    // ASAP / ALAP levels (unit latencies) -> priority = least slack to the end
    int32_t depth = 0;
    for (auto& o : ops) {
        for (int32_t d : o.deps)
            if (op_of[d] >= 0) o.level = std::max(o.level, ops[op_of[d]].level + 1);
        depth = std::max(depth, o.level);
    }
    {   // multiplications on the longest chain (what bounds the number of multiply steps)
        std::vector<int32_t> md(ops.size(), 0);
        int32_t              best = 0;
        for (size_t i = 0; i < ops.size(); i++) {
            int32_t m = 0;
            for (int32_t d : ops[i].deps)
                if (op_of[d] >= 0) m = std::max(m, md[op_of[d]]);
            md[i] = m + (ops[i].cls == CS_MUL ? 1 : 0);
            best  = std::max(best, md[i]);
        }
        P->mul_depth = (uint32_t)best;
        P->depth     = (uint32_t)depth;
    }
    for (auto& o : ops) o.alap = depth;
    for (int32_t i = (int32_t)ops.size() - 1; i >= 0; i--)
        for (int32_t d : ops[i].deps)
            if (op_of[d] >= 0) ops[op_of[d]].alap = std::min(ops[op_of[d]].alap, ops[i].alap - 1);
    // ---- list scheduling into steps of <= 64 operations of one class.  An operation is started no earlier than
    // `window` levels before its latest start (values are produced just in time: short lifetimes, small slot file).
    std::vector<std::vector<int32_t>> users(ops.size());
    std::vector<int32_t>              pending(ops.size(), 0);
    std::vector<int32_t>              left_at(depth + 2, 0); // unscheduled operations per latest-start level
    for (size_t i = 0; i < ops.size(); i++) {
        left_at[ops[i].alap]++;
        for (int32_t d : ops[i].deps)
            if (op_of[d] >= 0) {
                users[op_of[d]].push_back((int32_t)i);
                pending[i]++;
            }
    }
    std::vector<int32_t> ready[4];
    for (size_t i = 0; i < ops.size(); i++)
        if (pending[i] == 0) ready[ops[i].cls].push_back((int32_t)i);
    P->step_class.clear();
    std::vector<std::vector<int32_t>> step_ops;
    size_t                            done = 0;
    int32_t                           lvl = 0;     // least latest-start level among the unscheduled operations: those
    const int32_t                     window = COOP_WINDOW;  // operations are always ready (all their operands have smaller levels)
    while (done < ops.size()) {
        while (lvl <= depth && left_at[lvl] == 0) lvl++;
        bool any = false;
        for (uint8_t cls : {CS_LIN, CS_MUL, CS_INV}) {
            auto& R = ready[cls];
            if (R.empty()) continue;
            std::sort(R.begin(), R.end(), [&](int32_t x, int32_t y) {
                return ops[x].alap != ops[y].alap ? ops[x].alap < ops[y].alap : x < y;
            });
            // a step of this class is issued only when it is NEEDED (its most urgent operation is at its latest start);
            // operations with slack ride along in the lanes that step leaves free
            if (ops[R[0]].alap > lvl) continue;
            // a linear combination is evaluated by THREE lanes (three limbs each, see k_verify_coop), five groups per
            // 16-lane row (the carries move between neighbours with row-wise DPP shifts): 20 per step
            const size_t cap   = cls == CS_LIN ? 20 : 64;
            size_t       ntake = 0;
            while (ntake < R.size() && ntake < cap && ops[R[ntake]].alap <= lvl + window) ntake++;
            any = true;
            std::vector<int32_t> take(R.begin(), R.begin() + ntake);
            R.erase(R.begin(), R.begin() + ntake);
            const int32_t s = (int32_t)step_ops.size();
            for (int32_t x : take) {
                ops[x].step = s;
                left_at[ops[x].alap]--;
            }
            step_ops.push_back(take);
            P->step_class.push_back(cls);
            done += take.size();
            // operations whose last operand completed in this step become ready for the NEXT steps
            for (int32_t x : take)
                for (int32_t u : users[x])
                    if (--pending[u] == 0) ready[ops[u].cls].push_back(u);
        }
        if (!any) throw std::logic_error("coop program: scheduler stuck");
    }
    // ---- slots by liveness: a value's slot is free again after the last step that reads it
    const size_t         n_steps = step_ops.size();
    std::vector<int32_t> last_use(ops.size(), -1);
    for (size_t i = 0; i < ops.size(); i++)
        for (int32_t d : ops[i].deps)
            if (op_of[d] >= 0) last_use[op_of[d]] = std::max(last_use[op_of[d]], ops[i].step);
    for (int i = 0; i < 12; i++) last_use[op_of[out_node[i]]] = (int32_t)n_steps; // outputs live to the end
    std::vector<int32_t>              slot(ops.size(), -1);
    std::vector<int32_t>              free_slots;
    std::vector<std::vector<int32_t>> expire(n_steps + 2);
    uint32_t                          next_slot = P->in_base + COOP_N_INPUTS;
    auto node_slot = [&](int32_t v) -> uint32_t {
        if (B.nodes[v].op == t::N_IN) return (uint32_t)B.nodes[v].a;
        return (uint32_t)slot[op_of[v]];
    };
    P->words.assign(n_steps * 64, 0);
    P->terms.clear();
    for (size_t s = 0; s < n_steps; s++) {
        // (a slot whose last reader is step s may be written in step s: all lanes read before any lane writes)
        for (int32_t x : expire[s]) free_slots.push_back(slot[x]);
        uint32_t lane = 0;
        for (int32_t x : step_ops[s]) {
            int32_t sl;
            if (!free_slots.empty()) {
                sl = free_slots.back();
                free_slots.pop_back();
            } else
                sl = (int32_t)next_slot++;
            slot[x] = sl;
            if (last_use[x] < 0) throw std::logic_error("coop program: dead operation scheduled");
            if ((size_t)last_use[x] < n_steps) expire[std::max<size_t>(last_use[x], s + 1)].push_back(x);
            uint64_t w = (uint64_t)sl | (1ull << 63);
            const OpI& o = ops[x];
            if (o.cls == CS_MUL) {
                w |= (uint64_t)node_slot(o.deps[0]) << 14;
                w |= (uint64_t)node_slot(o.deps.size() > 1 ? o.deps[1] : o.deps[0]) << 28;
                P->n_mul_ops++;
            } else if (o.cls == CS_INV) {
                w |= (uint64_t)node_slot(o.deps[0]) << 14;
                P->n_inv_ops++;
            } else {
                const Comb& c = comb[o.node];
                w |= (uint64_t)c.size() << 14;
                w |= (uint64_t)P->terms.size() << 20;
                for (auto& kv : c) P->terms.push_back(((uint32_t)(uint16_t)(int16_t)kv.second << 16) | node_slot(kv.first));
                // padded to a whole number of trips with "0 x slot 0" (slot 0 is the constant zero): the kernel loads and
                // accumulates COOP_TRIP terms per trip without testing each
                while (P->terms.size() % COOP_TRIP) P->terms.push_back(0u);
                P->n_lin_ops++;
            }
            if (o.cls == CS_LIN) { // the same word for the three lanes of the group; lane 15 of every row stays idle
                const uint32_t l0 = (lane / 5) * 16 + (lane % 5) * 3; // `lane` counts groups here
                P->words[s * 64 + l0]     = w;
                P->words[s * 64 + l0 + 1] = w;
                P->words[s * 64 + l0 + 2] = w;
                lane++;
            } else
                P->words[s * 64 + lane++] = w;
        }
    }
    // a squaring is recorded as MUL(a, a): deps holds ONE entry then (handled above).  Slot numbers must fit 14 bits.
    P->n_slots = next_slot;
    if (next_slot >= (1u << 14) || P->terms.size() >= (1u << 24)) throw std::logic_error("coop program: encoding overflow");
    for (int i = 0; i < 12; i++) P->out_slot[i] = node_slot(out_node[i]);
}

// ------------------------------------------------------------------------------------------------ host interpreter
// Executes the program with the concrete field (tests; the device kernel is the same loop with lanes in parallel).
inline void coop_run_host(const CoopProgram& P, std::vector<Fq>& slots /* constants + inputs filled in */)
{
    slots.resize(P.n_slots, Fq::zero());
    const size_t n_steps = P.step_class.size();
    for (size_t s = 0; s < n_steps; s++) {
        Fq   res[64];
        bool val[64];
        for (int l = 0; l < 64; l++) {
            const uint64_t w = P.words[s * 64 + l];
            val[l]           = w >> 63;
            if (!val[l]) continue;
            const uint32_t a = (w >> 14) & 0x3fff, b = (w >> 28) & 0x3fff;
            if (P.step_class[s] == CS_MUL) {
                res[l] = fmul(slots[a], slots[b]);
            } else if (P.step_class[s] == CS_INV) {
                res[l] = finv(slots[a]);
            } else {
                const uint32_t nt = (w >> 14) & 0x3f, t0 = (uint32_t)((w >> 20) & 0xffffff);
                Fq             acc = Fq::zero();
                for (uint32_t k = 0; k < nt; k++) {
                    const uint32_t tw = P.terms[t0 + k];
                    int32_t        cf = (int16_t)(tw >> 16);
                    Fq             v  = slots[tw & 0xffff];
                    if (cf < 0) {
                        v  = fneg(v);
                        cf = -cf;
                    }
                    Fq add = Fq::zero(); // cf * v by double-and-add
                    for (int bit = 15; bit >= 0; bit--) {
                        add = fdbl(add);
                        if ((cf >> bit) & 1) add = fadd(add, v);
                    }
                    acc = fadd(acc, add);
                }
                res[l] = acc;
            }
        }
        for (int l = 0; l < 64; l++)
            if (val[l]) slots[P.words[s * 64 + l] & 0x3fff] = res[l];
    }
}

} // namespace k16
