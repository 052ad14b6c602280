// msm_g1.hip -- G1 (Fq) instantiation of the MSM kernels
#include <algorithm>
#include "msm_kernels.inc"
int k16_msm_enqueue_g1(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c)
{
    return msm_enqueue_t<k16::Fq>(ctx, (const k16::G1Aff*)d_bases, d_scalars, n, c);
}
