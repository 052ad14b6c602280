// msm_g2.hip -- G2 (Fq2) instantiation of the MSM kernels, canonical 8x32-bit field
#include <algorithm>
#include "msm_kernels.inc"
int k16_msm_enqueue_g2(k16_ctx* ctx, const void* d_bases, const void* d_scalars, uint64_t n, unsigned c, int prepared)
{
    (void)prepared; // G2 rows are used as they are
    return msm_enqueue_t<EngCanon<k16::Fq2>>(ctx, (const k16::G2Aff*)d_bases, d_scalars, n, c);
}
