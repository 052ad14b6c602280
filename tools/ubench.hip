// tools/ubench.hip -- instruction-rate microbenchmarks that steer the field-arithmetic design (dev aid).
// build: hipcc -O3 --offload-arch=gfx950 -I keyless-zk-proofs_amd/csrc tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "bn254_field.h"
#include "bn254_curve.h"
using namespace k16;

#define ITERS 2048
template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[8];
    double   d[8];
    for (int i = 0; i < 8; i++) { acc[i] = a * (i + 1); d[i] = (double)(a + i); }
    double dm = (double)b * 1e-9 + 1.0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            if (OP == 1) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 2) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 3) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
            if (OP == 5) { uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                           asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(b) : "vcc");
                           acc[i] = ((uint64_t)hi << 32) | lo; }
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm));
            if (OP == 7) { uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                           asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                           (void)lo; }
            if (OP == 8) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 9) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(*(uint32_t*)&acc[i]) : "v"(a), "v"(b)); }
        }
    }
    uint64_t s = 0; double ds = 0;
    for (int i = 0; i < 8; i++) { s += acc[i]; ds += d[i]; }
    if (s == 0x1234567 || ds == 1.2345) out[0] = 1;
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k_fmul_chain(Fq* out, const Fq* in, int iters)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fq  x = in[tid & 1023], y = in[(tid + 7) & 1023];
    for (int i = 0; i < iters; i++) {
        if (VARIANT == 0) { x = fmul(x, y); y = fmul(y, x); }
        if (VARIANT == 1) { x = fadd(x, y); y = fsub(y, x); }
    }
    out[tid & 1023] = fadd(x, y);
}
__global__ void __launch_bounds__(128) k_madd_chain(G1Xyzz* out, const G1Aff* in, int iters)
{
    int    tid = blockIdx.x * blockDim.x + threadIdx.x;
    G1Xyzz acc = G1Xyzz::zero();
    for (int i = 0; i < iters; i++) acc = padd_mixed(acc, in[(tid + i) & 1023]);
    out[tid & 1023] = acc;
}

// ---- radix-2^29 (9 limbs) Montgomery multiply feasibility: 64-bit column accumulators, no carries
struct F29 { uint32_t l[9]; };
__device__ __constant__ uint32_t P29[9];
__device__ __forceinline__ F29 fmul29(const F29& a, const F29& b, uint32_t np29)
{
    const uint32_t MASK = (1u << 29) - 1;
    uint32_t m[9];
    F29      r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a.l[i] * b.l[j];
            if (k < 9) { if (i < k) acc2 += (uint64_t)m[i] * P29[j]; }
            else acc2 += (uint64_t)m[i] * P29[j];
        }
        acc += acc2;
        if (k < 9) {
            m[k] = ((uint32_t)acc * np29) & MASK;
            acc += (uint64_t)m[k] * P29[0];
            acc >>= 29;
        } else {
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
__global__ void __launch_bounds__(256) k_fmul29_chain(F29* out, const F29* in, int iters, uint32_t np29)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    F29 x = in[tid & 1023], y = in[(tid + 7) & 1023];
    for (int i = 0; i < iters; i++) { x = fmul29(x, y, np29); y = fmul29(y, x, np29); }
    out[tid & 1023] = x;
}

template <class K, class... A>
float timeit(dim3 g, dim3 b, K k, A... a)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, 0, 0, a...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, g, b, 0, 0, a...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    uint32_t* d; hipMalloc(&d, 4096);
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32", "v_lshl_add_u64", "add_co+addc (2 instr)",
                           "v_fma_f64", "mad_u64+addc (2 instr)", "v_mul_u32_u24", "v_mad_u32_u24"};
    for (int waves = 1; waves <= 8; waves *= 2) {
        printf("--- %d waves/SIMD (blocks of 256 threads, %d blocks/CU)\n", waves, waves);
        dim3 g(256 * waves), b(256);
        float ms[10];
        ms[0] = timeit(g, b, k_rate<0>, d, 7u); ms[1] = timeit(g, b, k_rate<1>, d, 7u); ms[2] = timeit(g, b, k_rate<2>, d, 7u);
        ms[3] = timeit(g, b, k_rate<3>, d, 7u); ms[4] = timeit(g, b, k_rate<4>, d, 7u); ms[5] = timeit(g, b, k_rate<5>, d, 7u);
        ms[6] = timeit(g, b, k_rate<6>, d, 7u); ms[7] = timeit(g, b, k_rate<7>, d, 7u); ms[8] = timeit(g, b, k_rate<8>, d, 7u);
        ms[9] = timeit(g, b, k_rate<9>, d, 7u);
        for (int i = 0; i < 10; i++) {
            double ops = (double)256 * waves * 256 * ITERS * 8; // lane-ops (statement instances)
            printf("  %-24s %8.3f ms  %7.2f T lane-stmts/s  (%.2f per clk per SIMD @2.4GHz)\n", names[i], ms[i], ops / ms[i] / 1e9,
                   ops / (ms[i] * 1e-3) / 2.4e9 / 1024);
        }
    }
    Fq* in; Fq* out; hipMalloc(&in, 1024 * 64); hipMalloc(&out, 1024 * 128);
    hipMemset(in, 0x11, 1024 * 64);
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        dim3 g(256 * bpc), b(256);
        int  iters = 512;
        float m0 = timeit(g, b, k_fmul_chain<0>, out, (const Fq*)in, iters);
        float m1 = timeit(g, b, k_fmul_chain<1>, out, (const Fq*)in, iters);
        double nm = (double)256 * bpc * 256 * iters * 2;
        printf("fmul chain  %d blk/CU: %.3f ms -> %.1f G modmul/s ; add/sub chain %.3f ms -> %.1f G/s\n", bpc, m0, nm / m0 / 1e6, m1, nm / m1 / 1e6);
    }
    {
        // p in radix 2^29
        unsigned __int128 dummy = 0; (void)dummy;
        uint32_t p32[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        uint32_t p29[9];
        for (int i = 0; i < 9; i++) {
            int bit = 29 * i; uint64_t v = 0;
            for (int w = 0; w < 8; w++) { int off = 32 * w - bit; if (off > -32 && off < 29) v |= off >= 0 ? ((uint64_t)p32[w] << off) : ((uint64_t)p32[w] >> (-off)); }
            p29[i] = (uint32_t)(v & ((1u << 29) - 1));
        }
        hipMemcpyToSymbol(HIP_SYMBOL(P29), p29, sizeof p29);
        uint32_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - p29[0] * inv;
        uint32_t np29 = (0u - inv) & ((1u << 29) - 1);
        F29* in29; F29* out29; hipMalloc(&in29, 1024 * sizeof(F29)); hipMalloc(&out29, 1024 * sizeof(F29));
        hipMemset(in29, 0x05, 1024 * sizeof(F29));
        for (int bpc = 1; bpc <= 8; bpc *= 2) {
            dim3 g(256 * bpc), b(256);
            int iters = 512;
            float m0 = timeit(g, b, k_fmul29_chain, out29, (const F29*)in29, iters, np29);
            double nm = (double)256 * bpc * 256 * iters * 2;
            printf("fmul29 chain %d blk/CU: %.3f ms -> %.1f G modmul/s\n", bpc, m0, nm / m0 / 1e6);
        }
    }
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        dim3 g(256 * bpc), b(128);
        int  iters = 256;
        float m0 = timeit(g, b, k_madd_chain, (G1Xyzz*)out, (const G1Aff*)in, iters);
        double nm = (double)256 * bpc * 128 * iters;
        printf("madd chain  %d blk(128)/CU: %.3f ms -> %.2f G madd/s\n", bpc, m0, nm / m0 / 1e6);
    }
    return 0;
}
