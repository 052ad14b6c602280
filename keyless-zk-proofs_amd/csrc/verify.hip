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
#include "ctx.h"
#include "bn254_pairing.h"

using namespace k16;

struct k16_vk {
    k16_ctx*    ctx   = nullptr;
    uint32_t    n_ic  = 0;
    G1Aff*      d_ic  = nullptr; // IC[0 .. n_ic)
    G2Aff*      d_g2  = nullptr; // [0] -gamma, [1] -delta  (ark-groth16 PreparedVerifyingKey::gamma_g2_neg_pc / delta_g2_neg_pc)
    PairConsts* d_K   = nullptr;
    Fp12*       d_eab = nullptr; // e(alpha, beta)           (PreparedVerifyingKey::alpha_g1_beta_g2)
};

namespace {

// proof i: A (64 B) | B (128 B) | C (64 B), affine Montgomery.  Writes the three (P, Q) pairs of the check.
__global__ void __launch_bounds__(64) k_verify_prepare(const uint8_t* __restrict__ proofs, const uint8_t* __restrict__ inputs,
                                                       uint64_t n, uint32_t n_ic, const G1Aff* __restrict__ ic,
                                                       const G2Aff* __restrict__ neg_g2, G1Aff* __restrict__ P,
                                                       G2Aff* __restrict__ Q)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* pr = proofs + i * 256;
    G1Aff a, c;
    G2Aff b;
    memcpy(&a, pr, 64);
    memcpy(&b, pr + 64, 128);
    memcpy(&c, pr + 192, 64);
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
                                                   uint8_t* __restrict__ ok, Fp12* __restrict__ gt)
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
        ok[i]  = (good && f12_eq(e, t)) ? 1 : 0;
    }
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
                       const Fp12* d_target, uint8_t* d_ok, Fp12* d_gt, Fp12* d_f)
{
    hipStream_t st = ctx->stream;
    const uint64_t m = n * per;
    hipLaunchKernelGGL(k_pair_miller, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, d_P, d_Q, m, d_K, d_f);
    hipLaunchKernelGGL(k_pair_final, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_f, n, per, d_K, d_target, d_ok, d_gt);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}

} // namespace

extern "C" void k16_vk_destroy(k16_vk* vk)
{
    k16_guard_void([&]() {
    if (!vk) return;
    if (vk->ctx) (void)hipSetDevice(vk->ctx->device);
    void* bufs[] = {vk->d_ic, vk->d_g2, vk->d_K, vk->d_eab};
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
    *out = vk;
    return K16_OK;
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
    hipLaunchKernelGGL(k_verify_prepare, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_pr, d_in, n, vk->n_ic, vk->d_ic,
                       vk->d_g2, d_P, d_Q);
    int rc = pairings_on_device(ctx, vk->d_K, d_P, d_Q, n, 3, vk->d_eab, d_ok, nullptr, d_f);
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
