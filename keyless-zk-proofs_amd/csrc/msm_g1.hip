// msm_g1.hip -- G1 instantiation of the MSM kernels, on the radix-2^29 engine (bn254_fq9.h)
// one accumulator chain per product column in the G1 kernels (bn254_fq9.h K16_FMUL_CHAINS): 153 fewer 64-bit additions per mixed
// addition, 1 % on the bucket accumulation (round 5, profiles/r05/ab_accumulate_formulas.log); the NTT and G2 units keep two
#define K16_FMUL_CHAINS 1
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
    const void*       conv_in = nullptr;
    if (!prepared) {
        k16_ctx::Lane& L = ctx->lanes[ctx->cur_lane];
        int rc = k16_ws_reserve(ctx, L.ws_conv, (size_t)n * sizeof(k16::G1Aff));
        if (rc) return rc;
        rows = (const k16::G1Aff*)L.ws_conv.p;
        // the conversion rides in the sort's counting pass when there is one (LDS partition sort, no reused sort, no
        // captured graphs -- a graph would pin this call's table pointer); otherwise it is a kernel of its own
        const bool fuse = n <= (1u << 24) && !ctx->reuse_sort && ctx->derive_lane < 0 && !ctx->graphs_on && !ctx->tune.atomic_sort &&
                          !ctx->tune.no_fused_convert;
        if (ctx->lean_sort) {
            // lean sort: the counting pass stays at 25 VGPRs (the fused conversion made it 82), and the conversion is the
            // 5-doubling kernel -- both fit beside another lane's bucket accumulation
            hipLaunchKernelGGL(k_convert_bases_lean, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0,
                               k16_lane_stream(ctx, ctx->cur_lane), (const uint4*)d_bases, (uint4*)L.ws_conv.p, 2 * n);
            K16_HIP(ctx, hipGetLastError());
        } else if (fuse)
            conv_in = d_bases;
        else if ((rc = k16_msm_prepare_g1(ctx, d_bases, n, L.ws_conv.p, k16_lane_stream(ctx, ctx->cur_lane))))
            return rc;
    }
    return msm_enqueue_t<Eng9>(ctx, rows, d_scalars, n, c, false, conv_in);
}

// ---- fixed-base window tables: row w*n + i = 2^(c*w) * P_i in the packed R' layout.
// Table 0 is the prepared table itself; table w is c doublings of table w-1, normalised back to affine with one
// Fermat inversion per point (a load-time cost: ~n * W * (c + 400) multiplications).
namespace {
__global__ void __launch_bounds__(128) k_window_table_next(const k16::G1Aff* __restrict__ prev, k16::G1Aff* __restrict__ next,
                                                           uint64_t n, unsigned c)
{
    using namespace k16;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<Fq> w = load_vec(&prev[i]);
    Aff9    a{fq9_unpack(w.x.v), fq9_unpack(w.y.v)};
    Aff<Fq> o;
    for (int k = 0; k < 8; k++) o.x.v[k] = o.y.v[k] = 0; // (0,0) stays (0,0)
    if (!a.is_zero()) {
        Xyzz9 p = pdbl_aff9(a);
#pragma clang loop unroll(disable)
        for (unsigned k = 1; k < c; k++) p = pdbl9(p);
        if (!p.is_zero()) {
            // x = X / ZZ, y = Y / ZZZ with ONE inversion: t = 1 / (ZZ * ZZZ), 1/ZZ = t * ZZZ, 1/ZZZ = t * ZZ
            Fq9 t = finv9(fmul9(p.zz, p.zzz));
            Fq9 x = fmul9(p.x, fmul9(t, p.zzz)); // 8 * 2 -> < 2p
            Fq9 y = fmul9(p.y, fmul9(t, p.zz));  // 4 * 2 -> < 2p
            fq9_pack(o.x.v, x);
            fq9_pack(o.y.v, y);
        }
    }
    store_vec(&next[i], o);
}
} // namespace

// d_table: W * n rows; rows [0, n) receive the prepared form of d_bases (zkey format), the rest the window tables
int k16_msm_fixed_tables_g1(k16_ctx* ctx, const void* d_bases, uint64_t n, unsigned c, unsigned W, void* d_table)
{
    hipStream_t st = ctx->stream;
    int         rc = k16_msm_prepare_g1(ctx, d_bases, n, d_table, st);
    if (rc || n == 0) return rc;
    k16::G1Aff* t = (k16::G1Aff*)d_table;
    for (unsigned w = 1; w < W; w++)
        hipLaunchKernelGGL(k_window_table_next, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, t + (size_t)(w - 1) * n,
                           t + (size_t)w * n, n, c);
    K16_HIP(ctx, hipGetLastError());
    return K16_OK;
}
int k16_msm_enqueue_fixed_g1(k16_ctx* ctx, const void* d_table, const void* d_scalars, uint64_t n, unsigned c)
{
    return msm_enqueue_t<Eng9>(ctx, (const k16::G1Aff*)d_table, d_scalars, n, c, true);
}
int k16_msm_enqueue_classified_g1(k16_ctx* ctx, const void* d_rows, const k16_scalar_classes* cls, int set, unsigned c,
                                  bool* has_wide, int phase)
{
    return msm_enqueue_classified_t<Eng9>(ctx, (const k16::G1Aff*)d_rows, cls, set, c, has_wide, phase);
}
