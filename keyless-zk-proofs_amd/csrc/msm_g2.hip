// msm_g2.hip -- G2 (Fq2) instantiation of the MSM kernels
#include <algorithm>
#include "msm_kernels.inc"
int k16_msm_enqueue_g2(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c)
{
    return msm_enqueue_t<k16::Fq2>(ctx, (const k16::G2Aff*)d_bases, d_scalars, n, c);
}
