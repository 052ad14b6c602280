// msm_g1.hip -- G1 instantiation of the MSM kernels, on the radix-2^29 engine (bn254_fq9.h)
#include <algorithm>
#include "msm_kernels.inc"

// zkey-format table (canonical Montgomery, 64 B/point) -> packed R'-domain rows the kernels gather from
int k16_msm_prepare_g1(k16_ctx* ctx, const void* d_bases, uint64_t n, void* d_out, hipStream_t st)
{
    if (!st) st = ctx->stream;
    if (n == 0) return K16_OK;
    hipLaunchKernelGGL(k_convert_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const k16::G1Aff*)d_bases, (k16::G1Aff*)d_out, n);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}
int k16_msm_enqueue_g1(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c, int prepared)
{
    const k16::G1Aff* rows = (const k16::G1Aff*)d_bases;
    if (!prepared) {
        k16_ctx::Lane& L = ctx->lanes[ctx->cur_lane];
        int rc = k16_ws_reserve(ctx, L.ws_conv, (size_t)n * sizeof(k16::G1Aff));
        if (rc) return rc;
        if ((rc = k16_msm_prepare_g1(ctx, d_bases, n, L.ws_conv.p, L.stream))) return rc;
        rows = (const k16::G1Aff*)L.ws_conv.p;
    }
    return msm_enqueue_t<Eng9>(ctx, rows, d_scalars, n, c);
}
