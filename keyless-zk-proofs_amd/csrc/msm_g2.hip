// msm_g2.hip -- G2 instantiation of the MSM kernels: Fq2 over the radix-2^29 field (bn254_fq9.h, Fq2n)
#include <algorithm>
#include "msm_kernels.inc"

int k16_msm_prepare_g2(k16_ctx* ctx, const void* d_bases, uint64_t n, void* d_out, hipStream_t st)
{
    if (!st) st = ctx->stream;
    if (n == 0) return K16_OK;
    hipLaunchKernelGGL(k_convert_bases_g2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const k16::G2Aff*)d_bases, (k16::G2Aff*)d_out, n);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}
int k16_msm_enqueue_g2(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c, int prepared)
{
    const k16::G2Aff* rows = (const k16::G2Aff*)d_bases;
    if (!prepared) {
        k16_ctx::Lane& L = ctx->lanes[ctx->cur_lane];
        int rc = k16_ws_reserve(ctx, L.ws_conv, (size_t)n * sizeof(k16::G2Aff));
        if (rc) return rc;
        if ((rc = k16_msm_prepare_g2(ctx, d_bases, n, L.ws_conv.p, k16_lane_stream(ctx, ctx->cur_lane)))) return rc;
        rows = (const k16::G2Aff*)L.ws_conv.p;
    }
    return msm_enqueue_t<Eng2n>(ctx, rows, d_scalars, n, c);
}
int k16_msm_enqueue_classified_g2(k16_ctx* ctx, const void* d_rows, const k16_scalar_classes* cls, int set, unsigned c,
                                  bool* has_wide, int phase)
{
    return msm_enqueue_classified_t<Eng2n>(ctx, (const k16::G2Aff*)d_rows, cls, set, c, has_wide, phase);
}
