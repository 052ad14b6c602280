// verify.hip -- batched Groth16 verification on the GPU, behind the C ABI of include/k16.h.
//
// Replaces the CPU check the service runs on every proof before releasing it
// (prover-service/src/request_handler/prover_handler.rs:329-336: Groth16Proof::verify_proof -> ark-groth16 0.4.0
// prepare_inputs + verify_proof_with_prepared_inputs; types.rs:141-196 builds the prepared key):
//     e(A, B) * e(vk_x, -gamma) * e(C, -delta) == e(alpha, beta),      vk_x = IC[0] + sum_i x_i * IC[i+1]
// A batch of n proofs is n scalar-multiplication chains (vk_x), 3n independent Miller loops and n final
// exponentiations -- one lane each, three launches; e(alpha, beta) is computed once per key (k16_vk_create) by the same
// kernels.  The pairing arithmetic is csrc/bn254_pairing.h.
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include "ctx.h"
#include "bn254_pairing.h"
#include "bn254_fq9.h"
#include "verify_script.h"

using namespace k16;

struct k16_vk {
    k16_ctx*    ctx   = nullptr;
    uint32_t    n_ic  = 0;
    G1Aff*      d_ic  = nullptr; // IC[0 .. n_ic)
    G2Aff*      d_g2  = nullptr; // [0] -gamma, [1] -delta  (ark-groth16 PreparedVerifyingKey::gamma_g2_neg_pc / delta_g2_neg_pc)
    PairConsts* d_K   = nullptr;
    Fp12*       d_eab = nullptr; // e(alpha, beta)           (PreparedVerifyingKey::alpha_g1_beta_g2)
    // the wave-cooperative path (one wavefront per proof; verify_script.h): the program, this key's constant table
    // (curve constants, e(alpha, beta), line coefficients of -gamma and -delta) and 4-bit window tables of IC[1..]
    bool        coop      = false;
    uint64_t*   d_words   = nullptr;
    uint32_t *  d_terms = nullptr, *d_hdr = nullptr, *d_chunks = nullptr, *d_ctab9 = nullptr;
    G1Aff*      d_wtab    = nullptr; // [(n_ic - 1) * 64 windows][16 digits]: digit * 16^window * IC[j + 1]
    Fq*         d_target  = nullptr; // e(alpha, beta), 12 canonical values
    uint32_t    n_chunks = 0, chunk_words = 0, lds_bytes = 0;
    // device buffers of the small-batch (latency) case, allocated once: hipMalloc / hipFree per call cost more than 0.1 ms
    static constexpr uint64_t SMALL_N = 64;
    uint8_t *   d_small_pr = nullptr, *d_small_in = nullptr, *d_small_st = nullptr;
};

namespace {

// Input validation (what ark's deserialisation does before verify_proof ever sees a point; round-2 advisor finding): every
// coordinate must be a canonical field element (< p: otherwise A, A + p, A + 2p ... would be 2-3 encodings of one proof),
// and A, C must lie on y^2 = x^3 + 3, B on the twist y^2 = x^3 + 3 / (9 + u).  The all-zero encoding of the point at
// infinity passes (the pair then contributes 1).  A proof that fails is REJECTED (flag 0).  Membership of B in the
// r-torsion subgroup is still the caller's duty (include/k16.h).
__device__ __forceinline__ bool fq_canonical(const Fq& x)
{
    uint32_t borrow = 0; // x - p borrows  <=>  x < p
#pragma unroll
    for (int i = 0; i < 8; i++) borrow = (uint32_t)(((uint64_t)x.v[i] - FqParams::P[i] - borrow) >> 63);
    return borrow != 0;
}
__device__ __forceinline__ bool g1_input_ok(const G1Aff& a)
{
    if (!fq_canonical(a.x) || !fq_canonical(a.y)) return false;
    if (a.is_zero()) return true;
    const Fq three = fadd(fadd(Fq::one(), Fq::one()), Fq::one());
    return fsqr(a.y) == fadd(fmul(fsqr(a.x), a.x), three);
}
__device__ __forceinline__ bool g2_input_ok(const G2Aff& b, const Fq2& twist_b)
{
    if (!fq_canonical(b.x.a) || !fq_canonical(b.x.b) || !fq_canonical(b.y.a) || !fq_canonical(b.y.b)) return false;
    if (b.is_zero()) return true;
    return fsqr(b.y) == fadd(fmul(fsqr(b.x), b.x), twist_b);
}

// proof i: A (64 B) | B (128 B) | C (64 B), affine Montgomery.  Writes the three (P, Q) pairs of the check.
__global__ void __launch_bounds__(64) k_verify_prepare(const uint8_t* __restrict__ proofs, const uint8_t* __restrict__ inputs,
                                                       uint64_t n, uint32_t n_ic, const G1Aff* __restrict__ ic,
                                                       const G2Aff* __restrict__ neg_g2, G1Aff* __restrict__ P,
                                                       G2Aff* __restrict__ Q, const PairConsts* __restrict__ K,
                                                       uint8_t* __restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* pr = proofs + i * 256;
    G1Aff a, c;
    G2Aff b;
    memcpy(&a, pr, 64);
    memcpy(&b, pr + 64, 128);
    memcpy(&c, pr + 192, 64);
    bad[i] = (g1_input_ok(a) && g1_input_ok(c) && g2_input_ok(b, K->twist_b)) ? 0 : 1;
    // prepare_inputs (ark-groth16 verifier.rs): g_ic = IC[0] + sum_j x_j * IC[j + 1]; x_j is a 256-bit integer in standard
    // form (the service passes Fr::from_le_bytes_mod_order, i.e. any representative works: G1 has order r)
    G1Xyzz acc = G1Xyzz::from_aff(ic[0]);
#pragma clang loop unroll(disable)
    for (uint32_t j = 1; j < n_ic; j++) {
        uint8_t k[32];
        memcpy(k, inputs + (i * (n_ic - 1) + (j - 1)) * 32, 32);
        acc = padd(acc, pmul_scalar(G1Xyzz::from_aff(ic[j]), k));
    }
    P[3 * i + 0] = a;
    Q[3 * i + 0] = b;
    P[3 * i + 1] = to_affine(acc);
    Q[3 * i + 1] = neg_g2[0];
    P[3 * i + 2] = c;
    Q[3 * i + 2] = neg_g2[1];
}

__global__ void __launch_bounds__(64) k_pair_miller(const G1Aff* __restrict__ P, const G2Aff* __restrict__ Q, uint64_t m,
                                                    const PairConsts* __restrict__ K, Fp12* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    G1Aff p = P[i];
    G2Aff q = Q[i];
    PairConsts k = *K;
    Fp12 f;
    miller_loop(&f, &p, &q, &k);
    out[i] = f;
}

// lane i: product of `per` consecutive Miller-loop values, final exponentiation, comparison with *target (if given)
__global__ void __launch_bounds__(64) k_pair_final(const Fp12* __restrict__ f, uint64_t n, uint32_t per,
                                                   const PairConsts* __restrict__ K, const Fp12* __restrict__ target,
                                                   uint8_t* __restrict__ ok, Fp12* __restrict__ gt,
                                                   const uint8_t* __restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PairConsts k = *K;
    Fp12 acc = f[i * per];
#pragma clang loop unroll(disable)
    for (uint32_t j = 1; j < per; j++) {
        Fp12 t = f[i * per + j];
        f12_mul(&acc, &acc, &t);
    }
    Fp12 e;
    const bool good = final_exponentiation(&e, &acc, &k);
    if (gt) gt[i] = e;
    if (ok) {
        Fp12 t = *target;
        ok[i]  = (good && f12_eq(e, t) && !(bad && bad[i])) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// The wave-cooperative verifier: ONE wavefront checks one proof by interpreting the static program of verify_script.h.
// Slot file in LDS: field elements in the unsaturated radix-2^29 representation of bn254_fq9.h (9 limbs, values < 2p,
// R' = 2^261): a multiplication is fmul9, a linear combination is 64-bit limb-wise accumulation (no carries between
// terms) followed by ONE partial reduction.  Exact mod p throughout; the result is made canonical at the end.
struct CoopDev {
    const uint64_t* words;
    const uint32_t *terms, *hdr, *chunks;
    uint32_t        n_chunks, chunk_words, n_const, in_base, n_slots, target_const;
    uint32_t        out_slot[12];
};
constexpr uint32_t COOP_CHUNK_BYTES = 24576;

__device__ __forceinline__ Fq9 coop_ld9(const uint32_t* slots, uint32_t s)
{
    Fq9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = slots[s * 9 + k];
    return r;
}
__device__ __forceinline__ void coop_st9(uint32_t* slots, uint32_t s, const Fq9& v)
{
#pragma unroll
    for (int k = 0; k < 9; k++) slots[s * 9 + k] = v.l[k];
}
template <class T>
__device__ __forceinline__ T coop_shfl_down(const T& v, unsigned delta)
{
    static_assert(sizeof(T) % 4 == 0, "dword granularity");
    T               r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
    uint32_t*       d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) d[i] = (uint32_t)__shfl_down((int)s[i], delta, 64);
    return r;
}

// status[i]: 0 rejected, 1 accepted, 2 "not decided here" (vk_x is the point at infinity: the general path decides)
template <bool DBG>
__global__ void __launch_bounds__(128) k_verify_coop(CoopDev D, const uint32_t* __restrict__ ctab9, const G1Aff* __restrict__ wtab,
                                                    const G1Aff* __restrict__ ic, uint32_t n_ic,
                                                    const uint8_t* __restrict__ proofs, const uint8_t* __restrict__ inputs,
                                                    const Fq* __restrict__ target, uint8_t* __restrict__ status,
                                                    Fq* __restrict__ gt_out, uint64_t* __restrict__ dbg_ptr,
                                                    const Fq2* __restrict__ twist_b)
{
    uint64_t* const dbg = DBG ? dbg_ptr : nullptr; // (compile-time null in the production instantiation: no timing code)
    // dbg (K16_VERIFY_COOP_TRACE=1, proof 0 only): 100 MHz time stamps -- start, constants copied, vk_x done, inputs stored,
    // program done -- then the ticks spent staging chunks and in multiply / linear / inversion steps
    uint64_t tk[4] = {0, 0, 0, 0}, t_last = 0;
    auto     stamp = [&](int i) {
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    extern __shared__ uint32_t coop_lds[];
    // TWO wavefronts: wave 0 computes; wave 1 is the LOADER -- it stages chunk c + 1 of the program into the other half of
    // the staging buffer while wave 0 executes chunk c (the ~70 chunk loads, ~4 us each, used to sit between the chunks)
    uint32_t*      slots = coop_lds;
    uint32_t*      buf0  = coop_lds + ((D.n_slots * 9 + 3) & ~3u);
    const unsigned lane  = threadIdx.x & 63;
    const bool     loader = threadIdx.x >= 64;
    const uint64_t pi    = blockIdx.x;
    uint32_t*      ctab  = buf0 + 2 * D.chunk_words;
    auto stage_chunk = [&](uint32_t c) { // by the 64 lanes of ONE wave; chunk table already in LDS (or read from global for c = 0)
        const uint32_t s0 = D.chunks[4 * c], ns = D.chunks[4 * c + 1], tb = D.chunks[4 * c + 2], nt = D.chunks[4 * c + 3];
        uint32_t*      b  = buf0 + (c & 1) * D.chunk_words;
        const uint4*   wsrc = reinterpret_cast<const uint4*>(D.words + (size_t)s0 * 64);
        uint4*         wdst = reinterpret_cast<uint4*>(b);
        for (uint32_t k = lane; k < ns * 32; k += 64) wdst[k] = wsrc[k];
        for (uint32_t k = lane; k < nt; k += 64) b[ns * 128 + k] = D.terms[tb + k];
        uint32_t* hd = b + ns * 128 + ((nt + 3) & ~3u);
        for (uint32_t k = lane; k < ns; k += 64) hd[k] = D.hdr[s0 + k];
    };
    if (loader) {
        for (uint32_t k = lane; k < D.n_chunks * 4; k += 64) ctab[k] = D.chunks[k];
        stage_chunk(0);
    }
    uint32_t undecided = 0;
    if (!loader) {
    // ---- input validation (canonical coordinates, on the curve / the twist): an invalid proof is rejected here
    {
        bool ok_in = true;
        if (lane == 0 || lane == 1) {
            G1Aff a;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(proofs + pi * 256 + (lane == 0 ? 0 : 192));
#pragma unroll
            for (int k = 0; k < 8; k++) {
                a.x.v[k] = src[k];
                a.y.v[k] = src[8 + k];
            }
            ok_in = g1_input_ok(a);
        } else if (lane == 2) {
            G2Aff b;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(proofs + pi * 256 + 64);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                b.x.a.v[k] = src[k];
                b.x.b.v[k] = src[8 + k];
                b.y.a.v[k] = src[16 + k];
                b.y.b.v[k] = src[24 + k];
            }
            ok_in = g2_input_ok(b, *twist_b);
        }
        if (__ballot(ok_in) != ~0ull) undecided = 3; // decided: rejected
    }
    if (undecided == 0) {
    // ---- constants of the key -> slots [0, n_const)
    {
        const uint4* csrc = reinterpret_cast<const uint4*>(ctab9); // (table padded to a multiple of 4 words by the host)
        uint4*       cdst = reinterpret_cast<uint4*>(slots);
        for (uint32_t k = lane; k < (D.n_const * 9 + 3) / 4; k += 64) cdst[k] = csrc[k];
    }
    // ---- vk_x = IC[0] + sum_j x_j IC[j+1] (ark-groth16 prepare_inputs): one table row per (input, 4-bit window), a lane
    // per window, then a shuffle tree.  x_j acts as a 256-bit integer (G1 has order r), as in the general path.
    stamp(1);
    G1Xyzz acc = G1Xyzz::zero();
    for (uint32_t idx = lane; idx < (n_ic - 1) * 64; idx += 64) {
        const uint32_t j = idx >> 6, w = idx & 63;
        const uint8_t  byte = inputs[(pi * (n_ic - 1) + j) * 32 + (w >> 1)];
        const uint32_t d    = (w & 1) ? (byte >> 4) : (byte & 15);
        if (d) acc = padd_mixed(acc, wtab[(size_t)idx * 16 + d]);
    }
#pragma clang loop unroll(disable)
    for (unsigned d = 32; d >= 1; d >>= 1) {
        G1Xyzz o = coop_shfl_down(acc, d); // (lanes >= 64 - d read their own value: their sums are not used)
        acc      = padd(acc, o);
    }
    Fq       vk_s[3] = {Fq::zero(), Fq::zero(), Fq::zero()}; // X ZZZ, Y ZZ, ZZ ZZZ: vk_x stays projective (verify_script.h)
    if (lane == 0) {
        acc = padd_mixed(acc, ic[0]);
        if (acc.is_zero()) {
            undecided = 1;
        } else {
            vk_s[0] = fmul(acc.x, acc.zzz);
            vk_s[1] = fmul(acc.y, acc.zz);
            vk_s[2] = fmul(acc.zz, acc.zzz);
        }
    }
    undecided = (uint32_t)__shfl((int)undecided, 0, 64);
    if (undecided && lane == 0) status[pi] = 2;
    if (!undecided) {
    stamp(2);
    // ---- inputs -> slots: A.x A.y | B.x.a B.x.b B.y.a B.y.b | C.x C.y | vk_x as (X ZZZ, Y ZZ, ZZ ZZZ)
    {
        Fq v = Fq::zero();
        if (lane < 8) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(proofs + pi * 256 + lane * 32);
#pragma unroll
            for (int k = 0; k < 8; k++) v.v[k] = src[k];
        }
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t x = (uint32_t)__shfl((int)vk_s[j].v[k], 0, 64);
                if (lane == 8 + j) v.v[k] = x;
            }
        const Fq9 v9 = fq9_from_fq(v);
        if (lane < COOP_N_INPUTS) coop_st9(slots, D.in_base + lane, v9);
    }
    } // !undecided
    } // input valid
    if (undecided == 3 && lane == 0) status[pi] = 0;
    } // !loader
    // the loader learns whether the proof is decided here through LDS
    __shared__ uint32_t s_undecided;
    if (threadIdx.x == 0) s_undecided = undecided;
    __syncthreads();
    if (s_undecided) return; // both waves
    stamp(3);
    // ---- the program
    constexpr int64_t MASK = (1 << 29) - 1;
#pragma clang loop unroll(disable)
    for (uint32_t c = 0; c < D.n_chunks; c++) {
        if (loader) { // stage the next chunk into the other half, then meet the compute wave at the barrier
            if (c + 1 < D.n_chunks) stage_chunk(c + 1);
            __syncthreads();
            continue;
        }
        uint32_t* buf = buf0 + (c & 1) * D.chunk_words;
        const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctab[4 * c + 1]),
                       tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctab[4 * c + 2]),
                       nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctab[4 * c + 3]);
        // one wavefront computes: LDS operations of a wave complete in program order, so a value stored by one lane is seen
        // by the loads any lane issues later -- no barrier inside a chunk
        uint32_t* hdrs = buf + ns * 128 + ((nt + 3) & ~3u);
        if (dbg) t_last = __builtin_amdgcn_s_memrealtime();
        // the header and the instruction word of step s + 1 are loaded while step s executes (a lone wavefront has nobody to
        // hide an LDS round trip behind, and these two would head every step's dependency chain)
        uint32_t h_nx = hdrs[0], wlo_nx = buf[lane * 2], whi_nx = buf[lane * 2 + 1];
#pragma clang loop unroll(disable)
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t h     = (uint32_t)__builtin_amdgcn_readfirstlane((int)h_nx); // class | longest combination << 8
            const uint32_t cls   = h & 0xff;
            const uint32_t wlo = wlo_nx, whi = whi_nx;
            if (s + 1 < ns) {
                h_nx   = hdrs[s + 1];
                wlo_nx = buf[((s + 1) * 64 + lane) * 2];
                whi_nx = buf[((s + 1) * 64 + lane) * 2 + 1];
            }
            const bool     valid = whi >> 31;
            const uint32_t dst   = wlo & 0x3fff;
            Fq9            r;
            if (cls == CS_MUL) {
                const uint32_t a = valid ? (wlo >> 14) & 0x3fff : 0u, b = valid ? ((wlo >> 28) | (whi << 4)) & 0x3fff : 0u;
                r = fmul9(coop_ld9(slots, a), coop_ld9(slots, b));
            } else if (cls == CS_LIN) {
                // A linear combination is evaluated by THREE lanes: lane 3j + g accumulates limbs 3g .. 3g+2 of every term
                // (a third of the loads and multiply-adds per lane), the quotient by p comes from the top lane, and the
                // carries travel lane to lane once at the end.
                const uint32_t maxt = h >> 8;
                const uint32_t l16  = lane & 15;
                const uint32_t g    = l16 - 3 * ((l16 * 11) >> 5); // l16 % 3 for l16 < 16 (lane 15 of a row is idle)
                const uint32_t lb   = 3 * g;
                const uint32_t ntl  = valid ? (wlo >> 14) & 0x3f : 0u;
                const uint32_t t0   = ((wlo >> 20) | ((whi & 0xfff) << 12)) - tb;
                // limb-wise 64-bit accumulation on top of 2^15 p (keeps the total positive: the negative coefficients of a
                // combination sum to at most COOP_MAX_COEF = 2^12 times values < 5p: materialised linear atoms)
                const uint32_t p0 = lb == 0 ? Fq9C::P[0] : lb == 3 ? Fq9C::P[3] : Fq9C::P[6];
                const uint32_t p1 = lb == 0 ? Fq9C::P[1] : lb == 3 ? Fq9C::P[4] : Fq9C::P[7];
                const uint32_t p2 = lb == 0 ? Fq9C::P[2] : lb == 3 ? Fq9C::P[5] : Fq9C::P[8];
                int64_t        a3[3] = {(int64_t)p0 << 15, (int64_t)p1 << 15, (int64_t)p2 << 15};
                // COOP_TRIP terms per trip (the host pads every combination to a multiple of it with 0 x slot 0), all loads of a
                // trip issued before the first use: a lone wavefront has nobody to hide an LDS round trip behind, and a trip
                // has two dependent ones (term words, then slot limbs).  No per-term tests.
                const uint32_t ntp = (ntl + COOP_TRIP - 1) & ~(COOP_TRIP - 1);
                (void)maxt;
#pragma clang loop unroll(disable)
                for (uint32_t t = 0; t < ntp; t += COOP_TRIP) {
                    uint32_t tw[COOP_TRIP], lim[COOP_TRIP][3];
#pragma unroll
                    for (int u = 0; u < (int)COOP_TRIP; u++) tw[u] = buf[ns * 128 + t0 + t + u];
#pragma unroll
                    for (int u = 0; u < (int)COOP_TRIP; u++)
#pragma unroll
                        for (int k = 0; k < 3; k++) lim[u][k] = slots[(tw[u] & 0xffff) * 9 + lb + k];
#pragma unroll
                    for (int u = 0; u < (int)COOP_TRIP; u++) {
                        const int32_t cf = (int32_t)tw[u] >> 16;
#pragma unroll
                        for (int k = 0; k < 3; k++) a3[k] += (int64_t)cf * (int64_t)(int32_t)lim[u][k]; // v_mad_i64_i32
                    }
                }
                // the quotient by p, estimated from the two top accumulators (lane g = 2 holds limbs 6, 7, 8; p / 2^232 =
                // 3171406.3; 2^44 / 3171407 = 5547122.9) and taken low: the result is non-negative and below 5p -- fine
                // for fmul9, whose operand bounds may multiply to 128
                const uint64_t top_est = (uint64_t)(a3[2] + (a3[1] >> 29));                       // < 2^38
                int64_t        q       = (int64_t)(((top_est >> 11) * 5547123ull) >> 33) - 2;      // floor(top / 3171407) - 2 .. - 4
                q                      = q < 0 ? 0 : q;
                {   // lane g = 2 of the group has it: row_shl:1 / row_shl:2 bring it to lanes g = 1 / 0  (q < 2^17)
                    const int q2 = (int)(uint32_t)q;
                    const int s1 = __builtin_amdgcn_update_dpp(0, q2, 0x101, 0xf, 0xf, true);
                    const int s2 = __builtin_amdgcn_update_dpp(0, q2, 0x102, 0xf, 0xf, true);
                    q            = (int64_t)(uint32_t)(g == 2 ? q2 : g == 1 ? s1 : s2);
                }
                a3[0] -= q * (int64_t)p0;
                a3[1] -= q * (int64_t)p1;
                a3[2] -= q * (int64_t)p2;
                // carries: lane g = 0 first, then 1 (with 0's carry), then 2; the top limb (limb 8) keeps everything above
                int64_t  cout = 0;
                uint32_t o0 = 0, o1 = 0, o2 = 0;
#pragma unroll
                for (int round = 0; round < 3; round++) {
                    const uint32_t clo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)cout, 0x111, 0xf, 0xf, true); // row_shr:1
                    const uint32_t chi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)cout >> 32), 0x111, 0xf, 0xf, true);
                    if ((int)g == round) {
                        const int64_t cin = round == 0 ? 0 : (int64_t)(((uint64_t)chi << 32) | clo);
                        int64_t       t   = a3[0] + cin;
                        o0                = (uint32_t)(t & MASK);
                        t                 = a3[1] + (t >> 29);
                        o1                = (uint32_t)(t & MASK);
                        t                 = a3[2] + (t >> 29);
                        if (round < 2) {
                            o2   = (uint32_t)(t & MASK);
                            cout = t >> 29;
                        } else {
                            o2 = (uint32_t)t;
                        }
                    }
                }
                if (valid) {
                    slots[dst * 9 + lb]     = o0;
                    slots[dst * 9 + lb + 1] = o1;
                    slots[dst * 9 + lb + 2] = o2;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (dbg) {
                    const uint64_t now = __builtin_amdgcn_s_memrealtime();
                    tk[cls] += now - t_last;
                    t_last = now;
                }
                continue;
            } else { // CS_INV
                r = fq9_zero();
                // Fermat on the radix-2^29 field (254 squarings + ~127 multiplications at ~0.3 us each for a lone wave): faster
                // here than the binary-GCD inversion of the canonical field (finv_bgcd: ~1000 dependent 8-limb steps)
                if (valid) {
                    const Fq9 x = coop_ld9(slots, (wlo >> 14) & 0x3fff); // < 6p: 2 * 6 <= 128, the multiply's operand bound
                    r           = fq9_one();
#pragma clang loop unroll(disable)
                    for (int bit = 253; bit >= 0; bit--) { // x^(p - 2); p - 2 differs from p only in its lowest word
                        r = fsqr9(r);
                        const uint32_t ew = (bit >> 5) == 0 ? FqParams::P[0] - 2u
                                                            : (bit >> 5) == 1 ? FqParams::P[1] : (bit >> 5) == 2 ? FqParams::P[2]
                                                            : (bit >> 5) == 3 ? FqParams::P[3] : (bit >> 5) == 4 ? FqParams::P[4]
                                                            : (bit >> 5) == 5 ? FqParams::P[5] : (bit >> 5) == 6 ? FqParams::P[6]
                                                                                                                : FqParams::P[7];
                        if ((ew >> (bit & 31)) & 1u) r = fmul9(r, x);
                    }
                }
            }
            if (valid) coop_st9(slots, dst, r); // (every lane's operand loads precede this store in program order)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (dbg) {
                const uint64_t now = __builtin_amdgcn_s_memrealtime();
                tk[cls] += now - t_last;
                t_last = now;
            }
        }
        __syncthreads(); // the loader has staged chunk c + 1 meanwhile
    }
    if (loader) return;
    stamp(4);
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0)
        for (int k = 0; k < 4; k++) dbg[5 + k] = tk[k];
    // ---- the GT value, canonical, against e(alpha, beta)
    bool same = true;
    if (lane < 12) {
        const Fq v = fq9_to_fq(coop_ld9(slots, D.out_slot[lane]));
        if (gt_out) gt_out[pi * 12 + lane] = v;
        same = v == target[lane];
    }
    const uint64_t all = __ballot(same);
    if (lane == 0) status[pi] = all == ~0ull ? 1 : 0;
}

// the program is the same for every key: built once per process
const CoopProgram* coop_program()
{
    static std::once_flag once;
    static CoopProgram*   prog = nullptr;
    std::call_once(once, []() {
        try {
            PairConsts K;
            pairing_consts_init(&K);
            CoopProgram* p = new CoopProgram();
            coop_build_program(K, p);
            prog = p;
        } catch (...) {
            prog = nullptr;
        }
    });
    return prog;
}

struct DevBufs {
    std::vector<void*> p;
    ~DevBufs()
    {
        for (void* q : p)
            if (q) (void)hipFree(q);
    }
    hipError_t alloc(void** out, size_t bytes)
    {
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};

int pairings_on_device(k16_ctx* ctx, const PairConsts* d_K, const G1Aff* d_P, const G2Aff* d_Q, uint64_t n, uint32_t per,
                       const Fp12* d_target, uint8_t* d_ok, Fp12* d_gt, Fp12* d_f, const uint8_t* d_bad = nullptr)
{
    hipStream_t st = ctx->stream;
    const uint64_t m = n * per;
    hipLaunchKernelGGL(k_pair_miller, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, d_P, d_Q, m, d_K, d_f);
    hipLaunchKernelGGL(k_pair_final, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_f, n, per, d_K, d_target, d_ok, d_gt, d_bad);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}

} // namespace

extern "C" void k16_vk_destroy(k16_vk* vk)
{
    k16_guard_void([&]() {
    if (!vk) return;
    if (vk->ctx) (void)hipSetDevice(vk->ctx->device);
    void* bufs[] = {vk->d_ic, vk->d_g2, vk->d_K, vk->d_eab, vk->d_words, vk->d_terms, vk->d_hdr, vk->d_chunks, vk->d_ctab9,
                    vk->d_wtab, vk->d_target, vk->d_small_pr, vk->d_small_in, vk->d_small_st};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete vk;
    });
}

extern "C" int k16_vk_create(k16_ctx* ctx, const void* alpha1, const void* beta2, const void* gamma2, const void* delta2,
                             const void* ic, uint32_t n_ic, k16_vk** out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !alpha1 || !beta2 || !gamma2 || !delta2 || !ic || n_ic < 1 || !out) return K16_ERR_ARG;
    *out = nullptr;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    k16_vk* vk = new k16_vk();
    vk->ctx    = ctx;
    vk->n_ic   = n_ic;
    auto fail = [&](const char* what, hipError_t e) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        k16_vk_destroy(vk);
        return K16_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc((void**)&vk->d_ic, (size_t)n_ic * sizeof(G1Aff))) != hipSuccess) return fail("hipMalloc ic", e);
    if ((e = hipMalloc((void**)&vk->d_g2, 2 * sizeof(G2Aff))) != hipSuccess) return fail("hipMalloc g2", e);
    if ((e = hipMalloc((void**)&vk->d_K, sizeof(PairConsts))) != hipSuccess) return fail("hipMalloc consts", e);
    if ((e = hipMalloc((void**)&vk->d_eab, sizeof(Fp12))) != hipSuccess) return fail("hipMalloc eab", e);
    PairConsts K;
    pairing_consts_init(&K);
    G2Aff neg[2];
    memcpy(&neg[0], gamma2, sizeof(G2Aff));
    memcpy(&neg[1], delta2, sizeof(G2Aff));
    for (G2Aff& g : neg)
        if (!g.is_zero()) g.y = fneg(g.y);
    hipStream_t st = ctx->stream;
    if ((e = hipMemcpyAsync(vk->d_ic, ic, (size_t)n_ic * sizeof(G1Aff), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(vk->d_g2, neg, sizeof neg, hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(vk->d_K, &K, sizeof K, hipMemcpyHostToDevice, st)) != hipSuccess)
        return fail("hipMemcpyAsync vk", e);
    // alpha_g1_beta_g2 = e(alpha, beta), once per key
    DevBufs tmp;
    G1Aff*  d_p = nullptr;
    G2Aff*  d_q = nullptr;
    Fp12*   d_f = nullptr;
    if ((e = tmp.alloc((void**)&d_p, sizeof(G1Aff))) != hipSuccess || (e = tmp.alloc((void**)&d_q, sizeof(G2Aff))) != hipSuccess ||
        (e = tmp.alloc((void**)&d_f, sizeof(Fp12))) != hipSuccess)
        return fail("hipMalloc", e);
    if ((e = hipMemcpyAsync(d_p, alpha1, sizeof(G1Aff), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(d_q, beta2, sizeof(G2Aff), hipMemcpyHostToDevice, st)) != hipSuccess)
        return fail("hipMemcpyAsync alpha/beta", e);
    int rc = pairings_on_device(ctx, vk->d_K, d_p, d_q, 1, 1, nullptr, nullptr, vk->d_eab, d_f);
    if (rc) {
        k16_vk_destroy(vk);
        return rc;
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail("k16_vk_create", e);
    // ---- the wave-cooperative path (latency: one wavefront per proof).  Any failure here only leaves it switched off.
    if (!ctx->tune.verify_no_coop) {
        const CoopProgram* P = coop_program();
        Fp12               eab;
        if (P && hipMemcpy(&eab, vk->d_eab, sizeof eab, hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<Ell> l1, l2;
            coop_prepare_lines(neg[0], K, &l1);
            coop_prepare_lines(neg[1], K, &l2);
            std::vector<Fq> ctab;
            coop_const_table(K, eab, l1, l2, &ctab);
            std::vector<uint32_t> ctab9((ctab.size() * 9 + 3) & ~(size_t)3, 0u);
            for (size_t i = 0; i < ctab.size(); i++) {
                const Fq9 v = fq9_from_fq(ctab[i]);
                for (int k = 0; k < 9; k++) ctab9[i * 9 + k] = v.l[k];
            }
            // window tables: row (j * 64 + w) * 16 + d = d * 16^w * IC[j + 1]   (d = 0 unused)
            std::vector<G1Aff> wtab((size_t)(n_ic - 1) * 64 * 16);
            for (uint32_t j = 0; j + 1 < n_ic; j++) {
                G1Aff base_aff;
                memcpy(&base_aff, (const uint8_t*)ic + (size_t)(j + 1) * sizeof(G1Aff), sizeof(G1Aff));
                G1Xyzz base = G1Xyzz::from_aff(base_aff);
                for (uint32_t w = 0; w < 64; w++) {
                    G1Xyzz m = base;
                    for (uint32_t d = 1; d < 16; d++) {
                        wtab[((size_t)j * 64 + w) * 16 + d] = to_affine(m);
                        m = padd(m, base);
                    }
                    base = m; // 16 * base
                    wtab[((size_t)j * 64 + w) * 16] = G1Aff{Fq::zero(), Fq::zero()};
                }
            }
            // program -> device: words, terms, per-step header (class | longest combination << 8), chunk table
            const size_t          n_steps = P->step_class.size();
            std::vector<uint32_t> hdr(n_steps), chunks;
            std::vector<uint32_t> step_t0(n_steps, 0), step_nt(n_steps, 0);
            for (size_t sidx = 0; sidx < n_steps; sidx++) {
                uint32_t mx = 0, lo = 0xffffffffu, hi = 0;
                if (P->step_class[sidx] == CS_LIN)
                    for (int l = 0; l < 64; l++) {
                        const uint64_t w = P->words[sidx * 64 + l];
                        if (!(w >> 63)) continue;
                        const uint32_t nt = (uint32_t)((w >> 14) & 0x3f), t0 = (uint32_t)((w >> 20) & 0xffffff);
                        mx = std::max(mx, nt);
                        lo = std::min(lo, t0);
                        hi = std::max(hi, t0 + ((nt + COOP_TRIP - 1) & ~(COOP_TRIP - 1))); // (combinations are padded to whole trips)
                    }
                hdr[sidx]     = P->step_class[sidx] | (mx << 8);
                step_t0[sidx] = hi ? lo : 0;
                step_nt[sidx] = hi ? hi - lo : 0;
            }
            uint32_t max_chunk_words = 0;
            for (size_t s0 = 0; s0 < n_steps;) { // greedy: as many consecutive steps as fit the staging buffer
                size_t   s1 = s0;
                uint32_t tlo = 0, thi = 0;
                bool     have = false;
                while (s1 < n_steps) {
                    uint32_t nlo = tlo, nhi = thi;
                    bool     nh  = have;
                    if (step_nt[s1]) {
                        nlo = have ? std::min(tlo, step_t0[s1]) : step_t0[s1];
                        nhi = have ? std::max(thi, step_t0[s1] + step_nt[s1]) : step_t0[s1] + step_nt[s1];
                        nh  = true;
                    }
                    const size_t bytes = (s1 + 1 - s0) * 512 + (size_t)(nh ? nhi - nlo : 0) * 4;
                    if (bytes > COOP_CHUNK_BYTES && s1 > s0) break;
                    tlo = nlo, thi = nhi, have = nh;
                    s1++;
                }
                chunks.push_back((uint32_t)s0);
                chunks.push_back((uint32_t)(s1 - s0));
                chunks.push_back(have ? tlo : 0);
                chunks.push_back(have ? thi - tlo : 0);
                max_chunk_words = std::max<uint32_t>(max_chunk_words, (uint32_t)((s1 - s0) * 129 + (have ? thi - tlo : 0) + 8));
                s0 = s1;
            }
            vk->n_chunks  = (uint32_t)(chunks.size() / 4);
            vk->chunk_words = (max_chunk_words + 3) & ~3u;
            vk->lds_bytes   = (((P->n_slots * 9 + 3) & ~3u) + 2 * vk->chunk_words + vk->n_chunks * 4) * 4; // two staging halves
            const Fq2* ev = &eab.c0.c0;
            Fq         target[12];
            for (int i = 0; i < 6; i++) {
                target[2 * i]     = ev[i].a;
                target[2 * i + 1] = ev[i].b;
            }
            auto up = [&](void** d, const void* h, size_t bytes) {
                return hipMalloc(d, bytes ? bytes : 16) == hipSuccess && hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
            };
            hipFuncAttributes fa{};
            const size_t      static_lds = hipFuncGetAttributes(&fa, (const void*)k_verify_coop<false>) == hipSuccess ? fa.sharedSizeBytes : 4096;
            vk->coop = vk->lds_bytes + static_lds <= 160 * 1024 &&
                       up((void**)&vk->d_words, P->words.data(), P->words.size() * 8) &&
                       up((void**)&vk->d_terms, P->terms.data(), P->terms.size() * 4) &&
                       up((void**)&vk->d_hdr, hdr.data(), hdr.size() * 4) && up((void**)&vk->d_chunks, chunks.data(), chunks.size() * 4) &&
                       up((void**)&vk->d_ctab9, ctab9.data(), ctab9.size() * 4) &&
                       up((void**)&vk->d_wtab, wtab.data(), wtab.size() * sizeof(G1Aff)) &&
                       up((void**)&vk->d_target, target, sizeof target) &&
                       hipMalloc((void**)&vk->d_small_pr, k16_vk::SMALL_N * 256) == hipSuccess &&
                       hipMalloc((void**)&vk->d_small_in, std::max<size_t>(k16_vk::SMALL_N * (n_ic - 1) * 32, 16)) == hipSuccess &&
                       hipMalloc((void**)&vk->d_small_st, k16_vk::SMALL_N) == hipSuccess &&
                       hipFuncSetAttribute((const void*)k_verify_coop<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)vk->lds_bytes) == hipSuccess &&
                       hipFuncSetAttribute((const void*)k_verify_coop<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)vk->lds_bytes) == hipSuccess;
            if (!vk->coop) (void)hipGetLastError();
        }
    }
    *out = vk;
    return K16_OK;
    });
}

// returns K16_OK (h_ok filled), 1 (some proof undecided: caller takes the general path), or an error
static int verify_coop(k16_ctx* ctx, const k16_vk* vk, const void* h_proofs, const void* h_inputs, uint64_t n, uint8_t* h_ok,
                       void* h_gt)
{
    hipStream_t  st = ctx->stream;
    DevBufs      tmp;
    uint8_t *    d_pr = nullptr, *d_in = nullptr, *d_st = nullptr;
    Fq*          d_gt = nullptr;
    const size_t in_bytes = (size_t)n * (vk->n_ic - 1) * 32;
    if (n <= k16_vk::SMALL_N) { // the latency case: buffers of the key, no allocation
        d_pr = vk->d_small_pr;
        d_in = vk->d_small_in;
        d_st = vk->d_small_st;
    } else {
        K16_HIP(ctx, tmp.alloc((void**)&d_pr, (size_t)n * 256));
        K16_HIP(ctx, tmp.alloc((void**)&d_in, in_bytes));
        K16_HIP(ctx, tmp.alloc((void**)&d_st, n));
    }
    if (h_gt) K16_HIP(ctx, tmp.alloc((void**)&d_gt, (size_t)n * 12 * sizeof(Fq)));
    const CoopProgram* P = coop_program();
    CoopDev            D;
    D.words = vk->d_words;
    D.terms = vk->d_terms;
    D.hdr = vk->d_hdr;
    D.chunks = vk->d_chunks;
    D.n_chunks = vk->n_chunks;
    D.chunk_words = vk->chunk_words;
    D.n_const = P->n_const;
    D.in_base = P->in_base;
    D.n_slots = P->n_slots;
    D.target_const = P->target_const;
    for (int i = 0; i < 12; i++) D.out_slot[i] = P->out_slot[i];
    const bool        trace = ctx->tune.verify_coop_trace;
    uint64_t*         d_dbg = nullptr;
    if (trace) K16_HIP(ctx, tmp.alloc((void**)&d_dbg, 16 * 8));
    // The latency case (n <= 64, no GT output): inputs and flags go through the context's pinned, device-mapped staging
    // area (its last 64 KB) -- no copy commands at all, the call is one launch and one stream wait.
    const bool     mapped = !h_gt && n <= k16_vk::SMALL_N && in_bytes <= 32768;
    uint8_t*       hp     = (uint8_t*)ctx->pinned + (size_t)k16_ctx::PEND_SLOTS * k16_ctx::SLOT_BYTES;
    const uint8_t* dp     = (const uint8_t*)ctx->pinned_dev + (size_t)k16_ctx::PEND_SLOTS * k16_ctx::SLOT_BYTES;
    const uint8_t *k_pr = d_pr, *k_in = d_in;
    uint8_t*       k_st = d_st;
    // the staging area and the key's small buffers are shared by every caller of this context: one latency verification at
    // a time (two threads could otherwise read each other's flags), from the staging copy to the read-back
    std::unique_lock<std::mutex> staging_lock(ctx->verify_mu, std::defer_lock);
    if (n <= k16_vk::SMALL_N) staging_lock.lock();
    if (mapped) {
        memcpy(hp, h_proofs, (size_t)n * 256);
        if (in_bytes) memcpy(hp + 16384, h_inputs, in_bytes);
        k_pr = dp;
        k_in = dp + 16384;
        k_st = (uint8_t*)dp + 49152;
    } else {
        K16_HIP(ctx, hipMemcpyAsync(d_pr, h_proofs, (size_t)n * 256, hipMemcpyHostToDevice, st));
        if (in_bytes) K16_HIP(ctx, hipMemcpyAsync(d_in, h_inputs, in_bytes, hipMemcpyHostToDevice, st));
    }
    if (trace)
        hipLaunchKernelGGL(k_verify_coop<true>, dim3((unsigned)n), dim3(128), vk->lds_bytes, st, D, vk->d_ctab9, vk->d_wtab, vk->d_ic,
                           vk->n_ic, k_pr, k_in, vk->d_target, k_st, d_gt, d_dbg, &vk->d_K->twist_b);
    else
        hipLaunchKernelGGL(k_verify_coop<false>, dim3((unsigned)n), dim3(128), vk->lds_bytes, st, D, vk->d_ctab9, vk->d_wtab, vk->d_ic,
                           vk->n_ic, k_pr, k_in, vk->d_target, k_st, d_gt, d_dbg, &vk->d_K->twist_b);
    K16_HIP(ctx, hipGetLastError());
    if (trace) {
        uint64_t h[16];
        K16_HIP(ctx, hipMemcpy(h, d_dbg, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "[k16 coop] us: constants %.1f vk_x %.1f inputs %.1f program %.1f | chunk staging %.1f mul %.1f lin %.1f inv %.1f\n",
                (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0, h[5] / 100.0, h[6] / 100.0,
                h[7] / 100.0, h[8] / 100.0);
    }
    std::vector<uint8_t> stt(n);
    if (mapped) {
        K16_HIP(ctx, hipStreamSynchronize(st));
        memcpy(stt.data(), hp + 49152, n);
    } else {
        K16_HIP(ctx, hipMemcpyAsync(stt.data(), d_st, n, hipMemcpyDeviceToHost, st));
        if (h_gt) K16_HIP(ctx, hipMemcpyAsync(h_gt, d_gt, (size_t)n * 12 * sizeof(Fq), hipMemcpyDeviceToHost, st));
        K16_HIP(ctx, hipStreamSynchronize(st));
    }
    for (uint64_t i = 0; i < n; i++)
        if (stt[i] > 1) return 1;
    if (h_ok) memcpy(h_ok, stt.data(), n);
    return K16_OK;
}

// parity tests: the GT value e(A,B) e(vk_x,-gamma) e(C,-delta) of every proof as the wave-cooperative path computes it
// (12 x 32 B, Montgomery, c0.c0.a first); K16_ERR_ARG when that path is not available for this key or these inputs
extern "C" int k16_verify_coop_gt(k16_ctx* ctx, const k16_vk* vk, const void* h_proofs, const void* h_inputs, uint64_t n,
                                  void* h_out_gt)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !vk || vk->ctx != ctx || !vk->coop || !n || !h_proofs || !h_out_gt || (vk->n_ic > 1 && !h_inputs)) return K16_ERR_ARG;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    int rc = verify_coop(ctx, vk, h_proofs, h_inputs, n, nullptr, h_out_gt);
    return rc == 1 ? K16_ERR_ARG : rc;
    });
}

extern "C" int k16_verify_batch(k16_ctx* ctx, const k16_vk* vk, const void* h_proofs, const void* h_inputs, uint64_t n,
                                uint8_t* h_ok)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !vk || vk->ctx != ctx || (n && (!h_proofs || !h_ok)) || (n && vk->n_ic > 1 && !h_inputs)) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // Small batches -- the service's one proof after every prove(), a wave of 64 -- take the wave-cooperative path: one
    // wavefront per proof, ~1-2 ms whatever n is up to the number of CUs, instead of the ~45 ms one lane needs for a
    // pairing.  It records the generic case only: a proof with a zero point, or whose vk_x is the point at infinity, sends
    // the batch to the general path below (same flags: both compute the same GT value for every other proof).
    const uint64_t coop_max = ctx->tune.verify_coop_max;
    if (vk->coop && n <= coop_max) {
        bool generic = true;
        for (uint64_t i = 0; i < n && generic; i++) {
            const uint8_t* pr = (const uint8_t*)h_proofs + i * 256;
            auto zero = [](const uint8_t* p, size_t len) {
                for (size_t k = 0; k < len; k++)
                    if (p[k]) return false;
                return true;
            };
            generic = !zero(pr, 64) && !zero(pr + 64, 128) && !zero(pr + 192, 64);
        }
        if (generic) {
            int rc = verify_coop(ctx, vk, h_proofs, h_inputs, n, h_ok, nullptr);
            if (rc == K16_OK) return rc;
            // 1 = a proof was left undecided; an error (a launch that did not fit after all, an allocation): either way the
            // general path below decides the batch -- the cooperative path is an accelerator, never the only way
            if (rc != 1) (void)hipGetLastError();
        }
    }
    DevBufs     tmp;
    uint8_t *   d_pr = nullptr, *d_in = nullptr, *d_ok = nullptr;
    G1Aff*      d_P = nullptr;
    G2Aff*      d_Q = nullptr;
    Fp12*       d_f = nullptr;
    const size_t in_bytes = (size_t)n * (vk->n_ic - 1) * 32;
    K16_HIP(ctx, tmp.alloc((void**)&d_pr, (size_t)n * 256));
    K16_HIP(ctx, tmp.alloc((void**)&d_in, in_bytes));
    K16_HIP(ctx, tmp.alloc((void**)&d_ok, n));
    K16_HIP(ctx, tmp.alloc((void**)&d_P, (size_t)3 * n * sizeof(G1Aff)));
    K16_HIP(ctx, tmp.alloc((void**)&d_Q, (size_t)3 * n * sizeof(G2Aff)));
    K16_HIP(ctx, tmp.alloc((void**)&d_f, (size_t)3 * n * sizeof(Fp12)));
    K16_HIP(ctx, hipMemcpyAsync(d_pr, h_proofs, (size_t)n * 256, hipMemcpyHostToDevice, st));
    if (in_bytes) K16_HIP(ctx, hipMemcpyAsync(d_in, h_inputs, in_bytes, hipMemcpyHostToDevice, st));
    uint8_t* d_bad = nullptr;
    K16_HIP(ctx, tmp.alloc((void**)&d_bad, n));
    hipLaunchKernelGGL(k_verify_prepare, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_pr, d_in, n, vk->n_ic, vk->d_ic,
                       vk->d_g2, d_P, d_Q, vk->d_K, d_bad);
    int rc = pairings_on_device(ctx, vk->d_K, d_P, d_Q, n, 3, vk->d_eab, d_ok, nullptr, d_f, d_bad);
    if (rc) return rc;
    K16_HIP(ctx, hipMemcpyAsync(h_ok, d_ok, n, hipMemcpyDeviceToHost, st));
    K16_HIP(ctx, hipStreamSynchronize(st));
    return K16_OK;
    });
}

// parity tests: out[i] = e(P_i, Q_i) as ark-ec's Bn::pairing computes it (12 x 32 B per value, c0.c0.a first)
extern "C" int k16_pairing_vec(k16_ctx* ctx, const void* h_g1, const void* h_g2, uint64_t n, void* h_out_gt)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || (n && (!h_g1 || !h_g2 || !h_out_gt))) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBufs     tmp;
    G1Aff*      d_P = nullptr;
    G2Aff*      d_Q = nullptr;
    Fp12 *      d_f = nullptr, *d_gt = nullptr;
    PairConsts* d_K = nullptr;
    K16_HIP(ctx, tmp.alloc((void**)&d_P, n * sizeof(G1Aff)));
    K16_HIP(ctx, tmp.alloc((void**)&d_Q, n * sizeof(G2Aff)));
    K16_HIP(ctx, tmp.alloc((void**)&d_f, n * sizeof(Fp12)));
    K16_HIP(ctx, tmp.alloc((void**)&d_gt, n * sizeof(Fp12)));
    K16_HIP(ctx, tmp.alloc((void**)&d_K, sizeof(PairConsts)));
    PairConsts K;
    pairing_consts_init(&K);
    K16_HIP(ctx, hipMemcpyAsync(d_K, &K, sizeof K, hipMemcpyHostToDevice, st));
    K16_HIP(ctx, hipMemcpyAsync(d_P, h_g1, n * sizeof(G1Aff), hipMemcpyHostToDevice, st));
    K16_HIP(ctx, hipMemcpyAsync(d_Q, h_g2, n * sizeof(G2Aff), hipMemcpyHostToDevice, st));
    int rc = pairings_on_device(ctx, d_K, d_P, d_Q, n, 1, nullptr, nullptr, d_gt, d_f);
    if (rc) return rc;
    K16_HIP(ctx, hipMemcpyAsync(h_out_gt, d_gt, n * sizeof(Fp12), hipMemcpyDeviceToHost, st));
    K16_HIP(ctx, hipStreamSynchronize(st));
    return K16_OK;
    });
}
