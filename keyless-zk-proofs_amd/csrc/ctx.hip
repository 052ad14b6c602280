// ctx.hip -- context, device memory, timers, and the batch field / point kernels used by the
// parity tests of the device arithmetic (fq_raw_generic.cpp:12-233, curve.cpp:91-458).
#include <string.h>
#include "ctx.h"

using namespace k16;

int k16_ws_reserve(k16_ctx* ctx, k16_devbuf& b, size_t bytes)
{
    if (b.bytes >= bytes) return K16_OK;
    if (b.p) {
        K16_HIP(ctx, hipStreamSynchronize(ctx->stream));
        K16_HIP(ctx, hipFree(b.p));
        b.p     = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    K16_HIP(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    return K16_OK;
}

extern "C" int k16_ctx_create(int device, k16_ctx** out)
{
    if (!out) return K16_ERR_ARG;
    *out  = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return K16_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return K16_ERR_NO_DEVICE;
    k16_ctx* c = new k16_ctx();
    c->device  = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess ||
        hipEventCreate(&c->ks_a) != hipSuccess || hipEventCreate(&c->ks_b) != hipSuccess) {
        delete c;
        return K16_ERR_NO_DEVICE;
    }
    c->pinned_bytes = 1 << 16;
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return K16_ERR_NO_DEVICE;
    }
    *out = c;
    return K16_OK;
}

extern "C" void k16_ctx_destroy(k16_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    k16_devbuf* bufs[] = {&c->ws_counts, &c->ws_offsets, &c->ws_cursor, &c->ws_sorted, &c->ws_segoff,
                          &c->ws_segbucket, &c->ws_partial, &c->ws_big, &c->ws_misc, &c->ws_lvl_a,
                          &c->ws_lvl_b, &c->ws_lvl_c, &c->ws_lvl_d, &c->ws_scan};
    for (auto* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& kv : c->ntt_tables)
        if (kv.second.roots) (void)hipFree(kv.second.roots);
    if (c->pinned) (void)hipHostFree(c->pinned);
    (void)hipEventDestroy(c->ev_a);
    (void)hipEventDestroy(c->ev_b);
    (void)hipEventDestroy(c->ks_a);
    (void)hipEventDestroy(c->ks_b);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* k16_last_error(const k16_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" void*       k16_stream(k16_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int k16_sync(k16_ctx* c)
{
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipStreamSynchronize(c->stream));
    return K16_OK;
}
extern "C" int k16_dev_alloc(k16_ctx* c, size_t bytes, void** dptr)
{
    if (!c || !dptr) return K16_ERR_ARG;
    K16_HIP(c, hipSetDevice(c->device));
    K16_HIP(c, hipMalloc(dptr, bytes ? bytes : 16));
    return K16_OK;
}
extern "C" int k16_dev_free(k16_ctx* c, void* dptr)
{
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipStreamSynchronize(c->stream));
    K16_HIP(c, hipFree(dptr));
    return K16_OK;
}
extern "C" int k16_h2d(k16_ctx* c, void* d, const void* h, size_t bytes)
{
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    return K16_OK;
}
extern "C" int k16_d2h(k16_ctx* c, void* h, const void* d, size_t bytes)
{
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    return K16_OK;
}
extern "C" int k16_timer_start(k16_ctx* c)
{
    if (!c) return K16_ERR_ARG;
    K16_HIP(c, hipEventRecord(c->ev_a, c->stream));
    return K16_OK;
}
extern "C" int k16_timer_stop(k16_ctx* c, float* ms)
{
    if (!c || !ms) return K16_ERR_ARG;
    K16_HIP(c, hipEventRecord(c->ev_b, c->stream));
    K16_HIP(c, hipEventSynchronize(c->ev_b));
    K16_HIP(c, hipEventElapsedTime(ms, c->ev_a, c->ev_b));
    return K16_OK;
}
extern "C" int k16_kernel_stats_enable(k16_ctx* c, int on)
{
    if (!c) return K16_ERR_ARG;
    c->stats_on = on != 0;
    return K16_OK;
}
extern "C" int k16_kernel_stats_reset(k16_ctx* c)
{
    if (!c) return K16_ERR_ARG;
    c->stats.clear();
    return K16_OK;
}
extern "C" int k16_kernel_stats_get(k16_ctx* c, const char* name, uint64_t* launches, double* total_ms)
{
    if (!c || !name) return K16_ERR_ARG;
    auto it = c->stats.find(name);
    if (launches) *launches = it == c->stats.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == c->stats.end() ? 0.0 : it->second.total_ms;
    return K16_OK;
}

// ---------------------------------------------------------------- batch field / point kernels
template <class PR>
__global__ void k_field_op(int op, const Fp<PR>* __restrict__ a, const Fp<PR>* __restrict__ b, Fp<PR>* __restrict__ r,
                           uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<PR> x = a[i], y = b ? b[i] : Fp<PR>::zero(), z;
    switch (op) {
    case K16_OP_ADD: z = fadd(x, y); break;
    case K16_OP_SUB: z = fsub(x, y); break;
    case K16_OP_NEG: z = fneg(x); break;
    case K16_OP_MUL: z = fmul(x, y); break;
    case K16_OP_SQR: z = fsqr(x); break;
    case K16_OP_TOMONT: z = to_mont(x); break;
    case K16_OP_FROMMONT: z = from_mont(x); break;
    default: z = Fp<PR>::zero();
    }
    r[i] = z;
}
template <class F>
__global__ void k_point_op(int op, const Xyzz<F>* __restrict__ p1, const void* __restrict__ p2, Xyzz<F>* __restrict__ r,
                           uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz<F> a = p1[i], z;
    switch (op) {
    case K16_PT_ADD: z = padd(a, ((const Xyzz<F>*)p2)[i]); break;
    case K16_PT_MADD: z = padd_mixed(a, ((const Aff<F>*)p2)[i]); break;
    case K16_PT_DBL: z = pdbl(a); break;
    default: z = Xyzz<F>::zero();
    }
    r[i] = z;
}

extern "C" int k16_field_op_vec(k16_ctx* c, int field, int op, const void* h_a, const void* h_b, void* h_r, uint64_t n)
{
    if (!c || !h_a || !h_r || (field != K16_FQ && field != K16_FR)) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    void * da = nullptr, *db = nullptr, *dr = nullptr;
    size_t bytes = (size_t)n * 32;
    K16_HIP(c, hipMalloc(&da, bytes));
    K16_HIP(c, hipMalloc(&dr, bytes));
    K16_HIP(c, hipMemcpyAsync(da, h_a, bytes, hipMemcpyHostToDevice, c->stream));
    if (h_b) {
        K16_HIP(c, hipMalloc(&db, bytes));
        K16_HIP(c, hipMemcpyAsync(db, h_b, bytes, hipMemcpyHostToDevice, c->stream));
    }
    unsigned grid = (unsigned)((n + 255) / 256);
    if (field == K16_FQ)
        hipLaunchKernelGGL((k_field_op<FqParams>), dim3(grid), dim3(256), 0, c->stream, op, (const Fq*)da,
                           (const Fq*)db, (Fq*)dr, n);
    else
        hipLaunchKernelGGL((k_field_op<FrParams>), dim3(grid), dim3(256), 0, c->stream, op, (const Fr*)da,
                           (const Fr*)db, (Fr*)dr, n);
    K16_HIP(c, hipGetLastError());
    K16_HIP(c, hipMemcpyAsync(h_r, dr, bytes, hipMemcpyDeviceToHost, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(da);
    (void)hipFree(dr);
    if (db) (void)hipFree(db);
    return K16_OK;
}

extern "C" int k16_point_op_vec(k16_ctx* c, int group, int op, const void* h_p1, const void* h_p2, void* h_r,
                                uint64_t n)
{
    if (!c || !h_p1 || !h_r || (group != K16_G1 && group != K16_G2)) return K16_ERR_ARG;
    if (n == 0) return K16_OK;
    size_t xb  = group == K16_G1 ? sizeof(G1Xyzz) : sizeof(G2Xyzz);
    size_t ab  = group == K16_G1 ? sizeof(G1Aff) : sizeof(G2Aff);
    size_t p2b = (op == K16_PT_MADD) ? ab : xb;
    void * d1 = nullptr, *d2 = nullptr, *dr = nullptr;
    K16_HIP(c, hipMalloc(&d1, n * xb));
    K16_HIP(c, hipMalloc(&dr, n * xb));
    K16_HIP(c, hipMemcpyAsync(d1, h_p1, n * xb, hipMemcpyHostToDevice, c->stream));
    if (h_p2 && op != K16_PT_DBL) {
        K16_HIP(c, hipMalloc(&d2, n * p2b));
        K16_HIP(c, hipMemcpyAsync(d2, h_p2, n * p2b, hipMemcpyHostToDevice, c->stream));
    }
    unsigned grid = (unsigned)((n + 63) / 64);
    if (group == K16_G1)
        hipLaunchKernelGGL((k_point_op<Fq>), dim3(grid), dim3(64), 0, c->stream, op, (const G1Xyzz*)d1, d2,
                           (G1Xyzz*)dr, n);
    else
        hipLaunchKernelGGL((k_point_op<Fq2>), dim3(grid), dim3(64), 0, c->stream, op, (const G2Xyzz*)d1, d2,
                           (G2Xyzz*)dr, n);
    K16_HIP(c, hipGetLastError());
    K16_HIP(c, hipMemcpyAsync(h_r, dr, n * xb, hipMemcpyDeviceToHost, c->stream));
    K16_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d1);
    (void)hipFree(dr);
    if (d2) (void)hipFree(d2);
    return K16_OK;
}
