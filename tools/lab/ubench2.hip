// tools/lab/ubench2.hip -- round 5: issue cost of the NON-multiply instructions of the radix-2^29 product
// (64-bit shift against alignbit + shift, masks, selects, three-operand adds) and the dependent-chain behaviour of
// v_mad_u64_u32 at the occupancy of the bucket accumulation (3 waves / SIMD, 1 or 2 chains per wave).
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/ubench2.hip -o tools/lab/ubench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 2048
template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[8];
    uint32_t lo[8], hi[8];
    for (int i = 0; i < 8; i++) {
        acc[i] = (uint64_t)a * (i + 1) * 0x100000001ull;
        lo[i]  = a + i;
        hi[i]  = b + i;
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i]));
            if (OP == 2) asm volatile("v_alignbit_b32 %0, %1, %0, 29\n\tv_lshrrev_b32 %1, 29, %1" : "+v"(lo[i]), "+v"(hi[i]));
            if (OP == 3) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(lo[i]));
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(hi[i]) : "vcc");
            if (OP == 5) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(hi[i]), "v"(a));
            if (OP == 6) asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(lo[i]));
            if (OP == 7) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(hi[i]));
            if (OP == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
            if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
            if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
            if (OP == 11) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(lo[i]) : "v"(hi[i]));
            if (OP == 12) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i] + lo[i] + hi[i];
    if (s == 0x1234567) out[0] = 1;
}

// dependent chains of v_mad_u64_u32: CH independent accumulators per lane, each a chain
template <int CH>
__global__ void __launch_bounds__(64) k_chain(uint32_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CH];
    for (int i = 0; i < CH; i++) acc[i] = a * (i + 1);
    for (int it = 0; it < ITERS * 8 / CH; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; i++) s += acc[i];
    if (s == 0x1234567) out[0] = 1;
}

// the same with the occupancy FORCED: 256-thread workgroups (one wave per SIMD) that each ask for 160 KB / W of LDS, so
// exactly W workgroups are resident per CU (the unforced grids above leave the placement to the dispatcher)
template <int CH>
__global__ void __launch_bounds__(256) k_chain_lds(uint32_t* out, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CH];
    for (int i = 0; i < CH; i++) acc[i] = a * (i + 1);
    for (int it = 0; it < ITERS * 8 / CH; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; i++) s += acc[i];
    if (s == 0x1234567) { out[0] = 1; lds[threadIdx.x] = 1; }
}
// select variants: VOP2 with VCC, VOP3 with an SGPR pair, and the xor / and / add form of a conditional negation
template <int OP>
__global__ void __launch_bounds__(256) k_sel(uint32_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint32_t lo[8], hi[8];
    for (int i = 0; i < 8; i++) { lo[i] = a + i; hi[i] = b + i; }
    uint64_t m = (a & 1) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull;
    m = __builtin_amdgcn_readfirstlane((uint32_t)m) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32);
    uint32_t mask = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(hi[i]), "s"(m));
            if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %2" : "+v"(lo[i]) : "v"(mask), "v"(hi[i]));
            if (OP == 2) { asm volatile("s_nop 0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(hi[i]) : "vcc"); }
            if (OP == 3) lo[i] = (mask & 1u) ? hi[i] - lo[i] : lo[i] + it; // what the compiler makes of a select
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += lo[i] + hi[i];
    if (s == 0x1234567) out[0] = 1;
}

template <class K, class... A>
float timeit(dim3 g, dim3 b, K k, A... a)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, 0, 0, a...);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, g, b, 0, 0, a...);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 4096);
    const char* names[] = {"v_mad_u64_u32", "v_lshrrev_b64 29", "alignbit+lshr32 (2 instr)", "v_and_b32 literal", "v_cndmask_b32",
                           "v_add3_u32", "v_ashrrev_i32", "v_sub_u32", "v_lshl_add_u64", "v_mul_lo_u32", "v_add_u32", "v_lshl_add_u32",
                           "add_co+addc (2 instr)"};
    const int   NOP = 13;
    for (int waves = 1; waves <= 8; waves *= 2) {
        if (waves == 2) continue;
        printf("--- %d waves/SIMD (blocks of 256 threads, %d blocks/CU); cost relative to v_add_u32 in the last column\n", waves, waves);
        dim3  g(256 * waves), b(256);
        float ms[NOP];
        ms[0]  = timeit(g, b, k_rate<0>, d, 7u);
        ms[1]  = timeit(g, b, k_rate<1>, d, 7u);
        ms[2]  = timeit(g, b, k_rate<2>, d, 7u);
        ms[3]  = timeit(g, b, k_rate<3>, d, 7u);
        ms[4]  = timeit(g, b, k_rate<4>, d, 7u);
        ms[5]  = timeit(g, b, k_rate<5>, d, 7u);
        ms[6]  = timeit(g, b, k_rate<6>, d, 7u);
        ms[7]  = timeit(g, b, k_rate<7>, d, 7u);
        ms[8]  = timeit(g, b, k_rate<8>, d, 7u);
        ms[9]  = timeit(g, b, k_rate<9>, d, 7u);
        ms[10] = timeit(g, b, k_rate<10>, d, 7u);
        ms[11] = timeit(g, b, k_rate<11>, d, 7u);
        ms[12] = timeit(g, b, k_rate<12>, d, 7u);
        for (int i = 0; i < NOP; i++) {
            double ops = (double)256 * waves * 256 * ITERS * 8;
            printf("  %-28s %8.3f ms  %7.2f T lane-stmts/s   x%.2f\n", names[i], ms[i], ops / ms[i] / 1e9, ms[i] / ms[10]);
        }
    }
    // chains: W waves per SIMD (blocks of 64 threads: W*4 blocks per CU), CH chains per lane
    printf("--- dependent v_mad_u64_u32 chains: T lane-ops/s by (waves per SIMD, chains per lane)\n");
    for (int w = 1; w <= 4; w++) {
        dim3  g(256 * 4 * w), b(64);
        float m1 = timeit(g, b, k_chain<1>, d, 7u), m2 = timeit(g, b, k_chain<2>, d, 7u), m4 = timeit(g, b, k_chain<4>, d, 7u),
              m8 = timeit(g, b, k_chain<8>, d, 7u);
        double ops = (double)256 * 4 * w * 64 * ITERS * 8;
        printf("  %d waves/SIMD: 1 chain %6.2f   2 chains %6.2f   4 chains %6.2f   8 chains %6.2f\n", w, ops / m1 / 1e9, ops / m2 / 1e9,
               ops / m4 / 1e9, ops / m8 / 1e9);
    }
    printf("--- the same with W workgroups of 256 threads resident per CU (LDS-forced): T lane-ops/s\n");
    for (int w = 1; w <= 5; w++) {
        const unsigned lds = (160u * 1024u) / w - 512u;
        auto run = [&](auto kern) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kern, dim3(256 * w), dim3(256), lds, 0, d, 7u);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(256 * w), dim3(256), lds, 0, d, 7u);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            return best;
        };
        double ops = (double)256 * w * 256 * ITERS * 8;
        float m1 = run(k_chain_lds<1>), m2 = run(k_chain_lds<2>), m4 = run(k_chain_lds<4>), m8 = run(k_chain_lds<8>);
        printf("  %d waves/SIMD: 1 chain %6.2f   2 chains %6.2f   4 chains %6.2f   8 chains %6.2f\n", w, ops / m1 / 1e9, ops / m2 / 1e9,
               ops / m4 / 1e9, ops / m8 / 1e9);
    }
    printf("--- selects (4 waves/SIMD): ms for the same statement count; v_add_u32 reference first\n");
    {
        dim3 g(256 * 4), b(256);
        float r0 = timeit(g, b, k_rate<10>, d, 7u);
        float s0 = timeit(g, b, k_sel<0>, d, 7u), s1 = timeit(g, b, k_sel<1>, d, 7u), s2 = timeit(g, b, k_sel<2>, d, 7u), s3 = timeit(g, b, k_sel<3>, d, 7u);
        printf("  v_add_u32 %.3f | v_cndmask_b32_e64 (sgpr pair) %.3f x%.2f | xor+add (2 instr) %.3f x%.2f | s_nop + v_cndmask vcc %.3f x%.2f | C ternary %.3f x%.2f\n",
               r0, s0, s0 / r0, s1, s1 / r0, s2, s2 / r0, s3, s3 / r0);
    }
    return 0;
}
