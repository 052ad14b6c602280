// tools/lab/ubench_fma64.hip -- VERDICT r5 item 8: would a 52-bit-limb Montgomery product on the FP64 pipe beat the radix-2^29
// v_mad_u64_u32 product (bn254_fq9.h: 162 multiply-adds, 174.3 G modmul/s chip-wide)?
//
// The FP64 route (Emmart / Zheng / Weems, "Faster modular exponentiation using double precision floating point arithmetic on
// the GPU"): limbs are integers < 2^52 held in doubles; ONE 52 x 52 -> 104-bit limb product takes, in round-toward-zero mode,
//     hi = fma(x, y, 2^104)            mantissa of hi  = floor(x y / 2^52)
//     t  = (2^104 + 2^52) - hi
//     lo = fma(x, y, t)                mantissa of lo  = x y mod 2^52
// and two 64-bit INTEGER additions of the raw bit patterns into the column accumulators (all values of a kind share one
// exponent, so their bit patterns add like integers; the biases are taken off once per column).  Five instructions for 2704
// bit^2 of product, against ONE v_mad_u64_u32 for 29 x 29 = 841 bit^2: 541 against 841 bit^2 per issued instruction -- on
// paper the FP64 route LOSES by 1.55x as long as v_fma_f64 issues no faster than v_mad_u64_u32.  A 254-bit product + Montgomery
// reduction on 5 x 52-bit limbs is 25 + 25 + 5 limb products = 55 x 5 = 275 instructions + ~40 of carry handling, against
// 162 + ~60.
//
// This bench measures the two ingredient rates so that the paper figure rests on this chip's numbers, not on a data sheet:
//   mode 0  chains of the five-instruction FP64 limb product (25 independent (x, y) pairs per lane, like one 5 x 5 product)
//   mode 1  chains of v_mad_u64_u32 column sums (81 per "product", like one 9 x 9 product)
// and prints limb products / s, bit^2 / s and the modular products / s each would bound (/ 55 and / 162).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/ubench_fma64.hip -o tools/lab/ubench_fma64 && tools/lab/ubench_fma64
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__device__ __forceinline__ double   as_f64(uint64_t v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ uint64_t as_u64(double v) { return (uint64_t)__double_as_longlong(v); }

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int iters)
{
    if (MODE == 0) {
        // FP64 round mode -> toward zero for this wave (MODE register bits 3:2 = DP round mode, 3 = toward zero)
        __builtin_amdgcn_s_setreg((1 << 11) | (2 << 6) | 1 /* hwreg(HW_REG_MODE, 2, 2) */, 3);
        double x[5], y[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            x[i] = (double)(in[threadIdx.x * 10 + i] & ((1ull << 52) - 1));
            y[i] = (double)(in[threadIdx.x * 10 + 5 + i] & ((1ull << 52) - 1));
        }
        const double C1 = as_f64(0x4670000000000000ull);                    // 2^104
        const double C2 = as_f64(0x4670000000000000ull) + 4503599627370496.0; // 2^104 + 2^52
        uint64_t     col[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const double hi = __builtin_fma(x[i], y[j], C1);
                    const double t  = C2 - hi;
                    const double lo = __builtin_fma(x[i], y[j], t);
                    col[i + j + 1] += as_u64(hi);
                    col[i + j] += as_u64(lo);
                }
            // feed the result back so that the iterations depend on each other (a real product's limbs feed the next one)
#pragma unroll
            for (int i = 0; i < 5; i++) x[i] = as_f64((col[i] & ((1ull << 52) - 1)) | 0x4330000000000000ull) - 4503599627370496.0;
        }
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 10; k++) acc ^= col[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    } else {
        uint32_t a[9], b[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            a[i] = (uint32_t)in[threadIdx.x * 18 + i] & 0x1fffffffu;
            b[i] = (uint32_t)in[threadIdx.x * 18 + 9 + i] & 0x1fffffffu;
        }
        uint64_t acc = 0;
        for (int it = 0; it < iters; it++) {
            uint64_t col[17];
#pragma unroll
            for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
            for (int i = 0; i < 9; i++)
#pragma unroll
                for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)a[i] * b[j];
            uint64_t f = 0; // every bit of every column feeds the next iteration: none of the 81 multiply-adds can be narrowed
#pragma unroll
            for (int k = 0; k < 17; k++) f ^= col[k] + (col[k] >> 29);
#pragma unroll
            for (int i = 0; i < 9; i++) a[i] = (uint32_t)(f >> (3 * i)) & 0x1fffffffu;
            acc ^= f;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ a[0];
    }
}

template <int MODE>
static double run(const uint64_t* d_in, uint64_t* d_out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters / 8 + 1);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    const int cus = prop.multiProcessorCount, iters = 4000;
    uint64_t  h_in[256 * 18];
    uint64_t  s = 0x9E3779B97F4A7C15ull;
    for (auto& v : h_in) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        v = s;
    }
    uint64_t *d_in, *d_out;
    hipMalloc(&d_in, sizeof h_in);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (int wg_per_cu : {4, 8}) { // 4 or 8 waves per SIMD
        const int blocks = cus * wg_per_cu;
        hipMalloc(&d_out, (size_t)blocks * 256 * 8);
        const double lanes = (double)blocks * 256;
        const double t0 = run<0>(d_in, d_out, blocks, iters), t1 = run<1>(d_in, d_out, blocks, iters);
        const double fp_limb = lanes * iters * 25 / t0, mad = lanes * iters * 81 / t1;
        printf("%d workgroups/CU (%d CUs, %s)\n", wg_per_cu, cus, prop.gcnArchName);
        printf("  FP64 limb products (fma, sub, fma, 2 x add_u64): %8.2f T/s = %7.1f P bit^2/s -> bounds a 5x52-bit Montgomery product at %6.1f G modmul/s (/55)\n",
               fp_limb * 1e-12, fp_limb * 2704 * 1e-15, fp_limb / 55 * 1e-9);
        printf("  v_mad_u64_u32 column sums (29-bit limbs):        %8.2f T/s = %7.1f P bit^2/s -> bounds a 9x29-bit Montgomery product at %6.1f G modmul/s (/162)\n",
               mad * 1e-12, mad * 841 * 1e-15, mad / 162 * 1e-9);
        hipFree(d_out);
    }
    return 0;
}
