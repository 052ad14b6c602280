// tools/lab/ubench_shoup.hip -- what a twiddle product costs as a Montgomery product (frmul9: 81 + 81 multiply-adds + 9 mul_lo) and
// with Shoup's precomputed quotient (w' = floor(w 2^261 / r) beside w: low half of w x, top half of w' x from column 7 up, low
// half of q r: 45 + 53 + 45).  Chains of dependent products, four independent values per lane (a double stage's four), 256-lane
// workgroups, four waves per SIMD like the NTT passes.  Prints the final limbs so that the host can check  x_N = w^N x_0 (mod r).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I keyless-zk-proofs_amd/csrc -I include tools/lab/ubench_shoup.hip -o tools/lab/ubench_shoup
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "bn254_curve.h"
#include "bn254_fq9.h"
using namespace k16;

constexpr uint32_t M29 = 0x1fffffffu;
// low half: (a * b) mod 2^261, limbs normalised
__device__ __forceinline__ Fr9 low9(const Fr9& a, const Fr9& b)
{
    Fr9      r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        r.l[k] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    return r;
}
// about floor(a * b / 2^261): columns 7 .. 16 (the carries of columns 0 .. 6 are dropped: the result is low by at most 2)
__device__ __forceinline__ Fr9 high9(const Fr9& a, const Fr9& b)
{
    Fr9      r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a.l[i] * b.l[j];
        }
        if (k >= 9) r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
__device__ __forceinline__ Fr9 sub_mod261(const Fr9& a, const Fr9& b)
{
    Fr9     r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t d = (int32_t)a.l[i] - (int32_t)b.l[i] + c;
        r.l[i] = (uint32_t)d & M29;
        c      = d >> 29;
    }
    return r;
}
__device__ __forceinline__ Fr9 rmod()
{
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = Fr9C::P[i];
    return r;
}
__device__ __forceinline__ Fr9 mul_shoup(const Fr9& w, const Fr9& wq, const Fr9& x)
{
    const Fr9 q = high9(wq, x);
    return sub_mod261(low9(w, x), low9(q, rmod()));
}

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters)
{
    Fr9 w, wq, x[4];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        w.l[i]  = in[i];
        wq.l[i] = in[9 + i];
#pragma unroll
        for (int v = 0; v < 4; v++) x[v].l[i] = in[18 + 9 * v + i];
    }
    x[0].l[0] ^= (threadIdx.x & 1); // (keeps the compiler from hoisting across lanes; lane 0 of block 0 is the one checked)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 4; v++) x[v] = MODE == 0 ? frmul9(w, x[v]) : mul_shoup(w, wq, x[v]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int v = 0; v < 4; v++)
            for (int i = 0; i < 9; i++) out[9 * v + i] = x[v].l[i];
    if (x[0].l[8] == 0xdeadbeefu) out[100] = 1;
}

int main()
{
    // r, w = 5^((r-1)/2^20) would do; any w < r: here w = 0x1234...; w' = floor(w 2^261 / r) -- both computed by the caller (python) and
    // passed on the command line would be nicer; kept self-contained: read 54 limbs from stdin
    uint32_t h[54];
    for (int i = 0; i < 54; i++)
        if (scanf("%u", &h[i]) != 1) return 2;
    uint32_t *d_in, *d_out;
    hipMalloc(&d_in, sizeof h);
    hipMalloc(&d_out, 4096);
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 512, blocks = 256 * 4 * 8; // 4 workgroups per CU resident, 8 rounds
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
            else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double mults = (double)blocks * 256 * 4 * iters;
            if (rep == 2) printf("%s: %.3f ms, %.1f G products/s\n", mode == 0 ? "montgomery" : "shoup     ", ms, mults / ms / 1e6);
        }
        uint32_t o[36];
        hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
        printf("limbs %d:", mode);
        for (int i = 0; i < 36; i++) printf(" %u", o[i]);
        printf("\n");
    }
    return 0;
}
