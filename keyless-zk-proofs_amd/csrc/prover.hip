// prover.hip -- Groth16 prover for BN254 on one MI355X, behind the C ABI of include/k16.h.
//
// Replaces (rust-rapidsnark/rapidsnark/src/):
//   BinFile / ZKeyUtils::Header / WtnsUtils::Header ... binfile_utils.cpp:13-58, zkey_utils.hpp:49-87,
//                                                       wtns_utils.hpp:29-44 (same on-disk formats)
//   Groth16::makeProver / Prover::prove ............... groth16.cpp:18-39, 41-360
//   Proof::toJson + json::dump ........................ groth16.cpp:378-410, fullprover.cpp:246
//
// The zkey is parsed once and its coefficient and point sections are uploaded to HBM once
// (the reference uses them in place from the mmap).  prove() runs the 4 witness MSMs, the
// sparse A.w / B.w product, the 3 x (iNTT, coset shift, NTT) chain, the H scalars and the H MSM
// on the device; the O(1) blinding arithmetic (groth16.cpp:325-352), the affine conversion and
// the decimal JSON stay on the host, using the same field code (bn254_field.h) compiled for x86.
#include <fcntl.h>
#include <sys/random.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <sched.h>
#include <string>
#include <thread>
#include <vector>
#include "ctx.h"
#include "bn254_fq9.h"
#include "spmv_plan.h"

#ifndef K16_CHAIN_PRIO
// Wave priority of the polynomial chain's kernels.  3 (above everything) until round 6; the witness MSMs' short kernels run at 3
// too and their accumulations at 0.  With the chain at 1 it still wins against the accumulations it runs beside, but the
// witness MSMs' fold / weighted-sum tails -- whose end, not the chain's, is what the H accumulation's start waits for -- are no
// longer held up by NTT waves: p50 over 11 alternating runs on two boxes 5.35-5.57 (median 5.46) against 5.37-5.84 (5.67) ms, equal on
// a third, two provers unchanged (profiles/r06/ab_wave_priorities.log, ab_chain_priority_second_box.log).  -DK16_CHAIN_PRIO=n to compare.
#define K16_CHAIN_PRIO 1
#endif
using namespace k16;

namespace {

static const uint8_t BN254_R_LE[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9,
                                       0x79, 0x48, 0xe8, 0x33, 0x28, 0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45,
                                       0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

// ---------------------------------------------------------------- iden3 binfile
struct Section {
    const uint8_t* p    = nullptr;
    uint64_t       size = 0;
};
struct BinView {
    Section sec[16];
};
// binfile_utils.cpp:13-58; first occurrence of a section type is "sectionPos 0"
int parse_binfile(const uint8_t* base, size_t size, const char* type, uint32_t max_version, BinView* out)
{
    if (size < 12 || memcmp(base, type, 4) != 0) return K16_ERR_FORMAT;
    uint32_t version, nsec;
    memcpy(&version, base + 4, 4);
    memcpy(&nsec, base + 8, 4);
    if (version > max_version) return K16_ERR_FORMAT;
    size_t pos = 12;
    for (uint32_t i = 0; i < nsec; i++) {
        if (pos + 12 > size) return K16_ERR_FORMAT;
        uint32_t st;
        uint64_t ss;
        memcpy(&st, base + pos, 4);
        memcpy(&ss, base + pos + 4, 8);
        pos += 12;
        if (ss > size - pos) return K16_ERR_FORMAT;
        if (st < 16 && out->sec[st].p == nullptr) {
            out->sec[st].p    = base + pos;
            out->sec[st].size = ss;
        }
        pos += ss;
    }
    return K16_OK;
}

struct MappedFile {
    uint8_t* base = nullptr;
    size_t   size = 0;
    int      fd   = -1;
    int      open_ro(const char* path)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return K16_ERR_IO;
        struct stat sb;
        if (fstat(fd, &sb) < 0) return K16_ERR_IO;
        size = (size_t)sb.st_size;
        if (size == 0) return K16_ERR_FORMAT;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return K16_ERR_IO;
        base = (uint8_t*)m;
        return K16_OK;
    }
    ~MappedFile()
    {
        if (base) munmap(base, size);
        if (fd >= 0) ::close(fd);
    }
};

// ---------------------------------------------------------------- device kernels (F12, F13)
__device__ __forceinline__ Fr ld_fr(const Fr* p)
{
    Fr           r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint4        a = s[0], b = s[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void st_fr(Fr* p, const Fr& r)
{
    uint4* d = reinterpret_cast<uint4*>(p);
    d[0]     = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    d[1]     = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// The polynomial chain works on the radix-2^29 representation of Fr (bn254_fq9.h): a, b, c live in HBM as
// packed R' values (x * 2^261 mod r, < 2r, 32 bytes), the coefficients are stored pre-multiplied by 2^522
// (the reference's zkey stores them pre-multiplied by R^2 = 2^512 for the same reason), and only the
// final H scalars are brought back to the canonical standard form the MSM consumes.  Field values are exact
// mod r throughout, so the H scalars are bit-identical to the reference's (tests compare them).
__device__ __forceinline__ Fr9 ld_r9(const Fr* p)
{
    Fr w = ld_fr(p);
    return fr9_load(w.v);
}
__device__ __forceinline__ void st_r9(Fr* p, const Fr9& v)
{
    Fr w;
    fr9_store(w.v, v);
    st_fr(p, w);
}
// groth16.cpp:137-156 : ab[c] += wtns[s] (x) coef.  The zkey's coefficient list is regrouped once at load time into rows
// (matrix m, constraint c), so each output element has one owner and no 256-bit atomics / spinlocks are needed.  Field
// addition is exact, so the summation order is free.  Layout (k16_prover_create): rows of up to SPMV_LONG entries are
// sorted by length and packed 64 to a SLICE, entry k of lane l at slice.off + 64 k + l -- a wave reads 2 KB of
// coefficients and 256 B of wire indices per step, and all its lanes loop the same number of times; longer rows (a circom
// Num2Bits or a big linear combination: hundreds to thousands of entries) get a whole wave each, lanes striding over the
// row's contiguous entries, and a butterfly reduction.  Rows land at their bit-reversed position: the inverse transforms
// that follow skip their own reversal.
__device__ __forceinline__ void spmv_store(Fr* __restrict__ a, Fr* __restrict__ b, uint32_t row, uint32_t N, uint32_t logN,
                                           const Fr9& acc)
{
    const uint32_t r   = row < N ? row : row - N;
    const uint32_t pos = logN ? (__brev(r) >> (32 - logN)) : 0u;
    st_r9(row < N ? &a[pos] : &b[pos], acc);
}
__global__ void __launch_bounds__(256) k_spmv(const SpmvSlice* __restrict__ slices, uint32_t n_slices,
                                              const uint32_t* __restrict__ row_of, const SpmvLong* __restrict__ longs,
                                              uint32_t n_long, const uint32_t* __restrict__ wire,
                                              const Fr* __restrict__ coef9, const Fr* __restrict__ wtns,
                                              Fr* __restrict__ a, Fr* __restrict__ b, uint32_t N, uint32_t logN,
                                              const uint16_t* __restrict__ n16)
{
    __builtin_amdgcn_s_setprio(K16_CHAIN_PRIO); // the polynomial chain gates the H MSM: its waves win VALU arbitration beside the witness MSMs
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    Fr9            acc = fq9_zero();
    // n16 (round 4): one 16-bit word per wire -- the value when it is below 256, bit 15 when it is not.  98 % of a circuit's
    // wires are bits and bytes: the walk's dependent gather then hits a 2.7 MB array (L2) instead of the 43 MB witness, and
    // the product is a single-limb multiplication; only the wide wires load their 32 bytes.  Same integers, same limbs.
    auto term = [&](uint32_t e) -> Fr9 {
        const uint32_t wi = wire[e];
        if (n16) {
            const uint32_t c = n16[wi];
            if (!(c & 0x8000u)) return fmul9_small_t<Fr9C>(ld_r9(&coef9[e]), c);
        }
        return frmul9(ld_r9(&wtns[wi]), ld_r9(&coef9[e]));
    };
    if (w < n_slices) {
        const SpmvSlice sl = slices[w];
        for (uint32_t k = 0; k < sl.len; k++) {
            const uint32_t e = sl.off + (k << 6) + lane; // padding entries: wire 0, coefficient 0
            acc = fradd9(acc, term(e));
        }
        const uint32_t row = row_of[(w << 6) + lane];
        if (row != 0xffffffffu) spmv_store(a, b, row, N, logN, acc);
        return;
    }
    if (w - n_slices >= n_long) return;
    const SpmvLong L = longs[w - n_slices];
    for (uint32_t k = lane; k < L.len; k += 64) {
        const uint32_t e = L.off + k;
        acc = fradd9(acc, term(e));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Fr9 o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = (uint32_t)__shfl_xor((int)acc.l[i], d, 64);
        acc = fradd9(acc, o);
    }
    if (lane == 0) spmv_store(a, b, L.row, N, logN, acc);
}
// groth16.cpp:160-167
__global__ void __launch_bounds__(256) k_mul(Fr* __restrict__ c, const Fr* __restrict__ a, const Fr* __restrict__ b,
                                             uint32_t N)
{
    __builtin_amdgcn_s_setprio(K16_CHAIN_PRIO); // the polynomial chain gates the H MSM: its waves win VALU arbitration beside the witness MSMs
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) st_r9(&c[i], frmul9(ld_r9(&a[i]), ld_r9(&b[i])));
}
// groth16.cpp:266-275 : a = fromMontgomery(a*b - c)  (standard form, canonical: the H MSM's scalars)
__global__ void __launch_bounds__(256) k_hscalars(Fr* __restrict__ out, const Fr* __restrict__ a,
                                                  const Fr* __restrict__ b, const Fr* __restrict__ c, uint32_t N)
{
    __builtin_amdgcn_s_setprio(K16_CHAIN_PRIO); // the polynomial chain gates the H MSM: its waves win VALU arbitration beside the witness MSMs
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) st_fr(&out[i], fr9_to_standard(frsub9(frmul9(ld_r9(&a[i]), ld_r9(&b[i])), ld_r9(&c[i]))));
}

// Witness upload in compact form.  A circom witness is n x 32 bytes of which ~98 % of the values are below 256 (bits and
// bytes): 43 MB for the Keyless circuit, 0.78 ms of PCIe time during which the GPU has nothing to do.  The host (a few
// threads, witness_pack below) splits it into one byte per wire + a list of the wide values (2.3 MB); this kernel rebuilds
// the n x 32-byte array in HBM, reading both straight from pinned, device-mapped host memory.  Same bytes as the plain copy.
// A grid-stride loop over at most 1024 long-lived ONE-WAVE workgroups: beside another prover's bucket accumulation -- two-wave
// workgroups at three waves per SIMD, thousands of them pending -- a workgroup of four waves finds room on a CU only when the
// accumulation's list runs dry, and the one-wire-per-lane version of this kernel (5,200 workgroups of 256) took 1.2-1.4 ms there
// instead of its 38 us; as one-wave workgroups 0.2 ms (profiles/r05/two_provers_interleaving.log).  One wave rebuilds 256
// wires per step from ONE 256-byte read of the pinned array (a dword per lane, the next step's issued before this step's
// stores); the bytes reach the lanes that store them by cross-lane reads, so that every store instruction of a wave covers
// 2 KB of consecutive HBM.
__global__ void __launch_bounds__(64) k_wtns_expand_narrow(const uint32_t* __restrict__ narrow4, Fr* __restrict__ out, uint32_t n,
                                                            uint16_t* __restrict__ n16)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_chunks = (n + 255u) / 256u, n_words = (n + 3u) / 4u; // (the array is allocated in whole dwords)
    uint32_t       c   = wave;
    uint32_t       nxt = (c < n_chunks && c * 64u + lane < n_words) ? narrow4[c * 64u + lane] : 0u;
    for (; c < n_chunks; c += n_waves) {
        const uint32_t cur = nxt;
        const uint32_t c2  = c + n_waves;
        if (c2 < n_chunks && c2 * 64u + lane < n_words) nxt = narrow4[c2 * 64u + lane];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t v = (uint32_t)__shfl((int)cur, (int)(k * 16u + (lane >> 2)));
            const uint32_t b = (v >> (8u * (lane & 3u))) & 0xffu;
            const uint32_t i = c * 256u + k * 64u + lane;
            if (i < n) {
                uint4* d = reinterpret_cast<uint4*>(&out[i]);
                d[0]     = make_uint4(b, 0u, 0u, 0u);
                d[1]     = make_uint4(0u, 0u, 0u, 0u);
                n16[i]   = (uint16_t)b; // (k_wtns_expand_wide, behind this kernel on the stream, flags the wide wires)
            }
        }
    }
}
// plain-copy path: the same 16-bit array from the full witness
__global__ void __launch_bounds__(256) k_wtns_n16(const uint4* __restrict__ wtns, uint32_t n, uint16_t* __restrict__ n16)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 lo = wtns[2 * (size_t)i], hi = wtns[2 * (size_t)i + 1];
    const bool  wide = ((lo.x >> 8) | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0;
    n16[i] = wide ? (uint16_t)0x8000u : (uint16_t)lo.x;
}
struct WideLists {
    const uint32_t* idx[32];
    const uint4*    val[32]; // 2 x uint4 per value
    uint32_t        count[32];
    uint32_t        n_lists;
    uint32_t        n_vars;
    uint32_t*       bad; // pinned, device-mapped: set when an entry names no wire of the circuit, or a wire whose byte is not 0
};
// (the lists are the host scan's -- or, through k16_prover_prove_compact, the CALLER's: an entry is checked before it is used,
// a bad one is skipped and reported when the proof's device work has been joined)
__global__ void __launch_bounds__(256) k_wtns_expand_wide(WideLists L, Fr* __restrict__ out, uint16_t* __restrict__ n16)
{
    const uint32_t t = blockIdx.y;
    if (t >= L.n_lists) return;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < L.count[t]; j += gridDim.x * blockDim.x) {
        const uint32_t i = L.idx[t][j];
        if (i >= L.n_vars) {
            *(volatile uint32_t*)L.bad = 1u; // (a plain store: no PCIe atomic needed; any bad entry's code will do)
            continue;
        }
        // claim the wire: bit 15 of its 16-bit entry, set atomically on the aligned word that holds it -- what was there before
        // says whether the wire had a byte of its own (code 2) or was listed already (code 4: a duplicate entry, round 6;
        // without the atomic two lanes could both find the entry free and the value that survives would depend on their order)
        const uint32_t sh  = (i & 1u) * 16u;
        const uint32_t old = (atomicOr(reinterpret_cast<uint32_t*>(n16) + (i >> 1), 0x8000u << sh) >> sh) & 0xffffu;
        if (old != 0) {
            *(volatile uint32_t*)L.bad = (old & 0x8000u) ? 4u : 2u;
            continue;
        }
        uint4*         d = reinterpret_cast<uint4*>(&out[i]);
        d[0]     = L.val[t][2 * j];
        d[1]     = L.val[t][2 * j + 1];
    }
}

// ---------------------------------------------------------------- host helpers
// fq.cpp:225-236 : fromMontgomery then base 10
std::string fq_to_dec(const Fq& m)
{
    Fq       s = from_mont(m);
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = s.v[i];
    uint32_t chunks[10];
    int      nch = 0;
    while (true) {
        uint32_t any = 0;
        for (int i = 0; i < 8; i++) any |= w[i];
        if (!any) break;
        uint64_t rem = 0;
        for (int i = 7; i >= 0; i--) {
            uint64_t cur = (rem << 32) | w[i];
            w[i]         = (uint32_t)(cur / 1000000000u);
            rem          = cur % 1000000000u;
        }
        chunks[nch++] = (uint32_t)rem;
    }
    if (nch == 0) return "0";
    char buf[100];
    int  len = snprintf(buf, sizeof buf, "%u", chunks[nch - 1]);
    for (int i = nch - 2; i >= 0; i--) len += snprintf(buf + len, sizeof buf - len, "%09u", chunks[i]);
    return std::string(buf, len);
}

bool geq_r(const uint8_t v[32])
{
    for (int i = 31; i >= 0; i--) {
        if (v[i] != BN254_R_LE[i]) return v[i] > BN254_R_LE[i];
    }
    return true;
}
// groth16.cpp:296-316 + random_generator.hpp:4-25 : 32 random bytes, top two bits cleared, rejected while >= r.  The bytes
// come from getrandom(2) (the kernel's CSPRNG, no file descriptor, no failure once the pool is initialised); a short or
// interrupted read is retried, and only a kernel without the system call falls back to /dev/urandom.
static bool random_bytes(uint8_t* out, size_t n)
{
    size_t got = 0;
    while (got < n) {
        const ssize_t r = ::getrandom(out + got, n - got, 0);
        if (r > 0) {
            got += (size_t)r;
            continue;
        }
        if (r < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        break; // ENOSYS and the like
    }
    if (got == n) return true;
    int fd = ::open("/dev/urandom", O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    while (got < n) {
        const ssize_t r = ::read(fd, out + got, n - got);
        if (r > 0) {
            got += (size_t)r;
            continue;
        }
        if (r < 0 && errno == EINTR) continue;
        break;
    }
    ::close(fd);
    return got == n;
}
int sample_blinding(uint8_t out[32])
{
    do {
        if (!random_bytes(out, 32)) return K16_ERR_IO;
        out[31] &= 0x3f;
    } while (geq_r(out));
    return K16_OK;
}
// the blinding scalars (and values derived from them) do not outlive the proof: wiped on every way out of the function
struct WipeOnExit {
    void*  p;
    size_t n;
    ~WipeOnExit() { explicit_bzero(p, n); }
};

int msm_prepared(k16_ctx* ctx, int group, const void* d_rows, const void* d_scalars, uint64_t n, void* out_xyzz)
{
    int rc = k16_msm_enqueue_prepared(ctx, group, d_rows, d_scalars, n);
    if (rc) return rc;
    return k16_msm_finish(ctx, out_xyzz, nullptr);
}

// out-of-line host group operations (keeps the host compile of this file short)
__attribute__((noinline)) G1Xyzz h_add(const G1Xyzz& a, const G1Xyzz& b) { return padd(a, b); }
__attribute__((noinline)) G1Xyzz h_madd(const G1Xyzz& a, const G1Aff& b) { return padd_mixed(a, b); }
__attribute__((noinline)) G1Xyzz h_mul(const G1Xyzz& a, const uint8_t* k) { return pmul_scalar(a, k); }
__attribute__((noinline)) G2Xyzz h_add(const G2Xyzz& a, const G2Xyzz& b) { return padd(a, b); }
__attribute__((noinline)) G2Xyzz h_madd(const G2Xyzz& a, const G2Aff& b) { return padd_mixed(a, b); }
__attribute__((noinline)) G2Xyzz h_mul(const G2Xyzz& a, const uint8_t* k) { return pmul_scalar(a, k); }

} // namespace

// The device-resident READ-ONLY part of a prover -- point tables in the kernels' row layout, the H window tables, the regrouped
// coefficients, the zero-row masks, the coset-shift table: 2.7 GB at the Keyless shape -- owned by reference count, so that
// several provers of one key on ONE device (FullProver's K16_DEVICES=0,0: throughput mode) upload and prepare it once
// (k16_prover_create_shared).  Freed on its device when the last prover that uses it goes.
struct ProverKeyOwner {
    int                device = 0;
    std::vector<void*> bufs;
    ~ProverKeyOwner()
    {
        (void)hipSetDevice(device);
        for (void* b : bufs)
            if (b) (void)hipFree(b);
    }
};

struct k16_prover {
    k16_ctx* ctx = nullptr;
    std::shared_ptr<ProverKeyOwner> key; // set once the key part is complete; until then prover_free frees the buffers itself
    uint32_t n_vars = 0, n_public = 0, domain_size = 0, logN = 0;
    uint64_t n_coefs = 0;
    G1Aff    alpha1, beta1, delta1;
    G2Aff    beta2, delta2;
    // device-resident key
    G1Aff *   d_A = nullptr, *d_B1 = nullptr, *d_C = nullptr, *d_H = nullptr;
    G1Aff*    d_Htab = nullptr; // fixed-base window tables of the H points (k16_msm_fixed_base_prepare), when available
    G2Aff*    d_B2    = nullptr;
    // constraint matrices A | B as length-sorted slices + long rows (see k_spmv)
    SpmvSlice* d_slices = nullptr;
    SpmvLong*  d_longs  = nullptr;
    uint32_t * d_rowof = nullptr, *d_wire = nullptr;
    Fr*        d_coef  = nullptr;
    uint32_t   n_slices = 0, n_long = 0;
    // per-proof buffers
    Fr *d_wtns = nullptr, *d_a = nullptr, *d_b = nullptr, *d_c = nullptr, *d_t[3] = {nullptr, nullptr, nullptr};
    uint16_t* d_n16 = nullptr; // per wire: the value when below 256, bit 15 otherwise (k_spmv)
    Fr* d_shift9 = nullptr; // 2^-k * g^i: between the inverse and the coset-forward transform (k16_ntt_build_coset_shift)
    k16_ntt_table* ntt = nullptr;
    hipStream_t    st2 = nullptr;          // polynomial chain (SpMV, NTTs) runs beside the witness MSMs
    hipEvent_t     ev_w = nullptr, ev_h = nullptr;
    std::vector<uint8_t> last_h;
    int warmup_rc = 0; // status of the create-time warm-up proof (k16_prover_warmup_status)
    struct WitnessPacker* packer = nullptr; // compact witness upload (see k_wtns_expand_*); null: plain copy
    // scalar classes of the witness (msm_classes.hip): one classification per proof serves A, B1, B2 and C.  zmask[t] = the
    // (0,0) rows of table t (A, B1, B2, C); tables with equal masks share a list set (set_of[t])
    k16_scalar_classes* cls = nullptr;
    void*               d_zmask[4] = {nullptr, nullptr, nullptr, nullptr};
    const void*         set_mask[4] = {nullptr, nullptr, nullptr, nullptr};
    int                 set_of[4] = {0, 0, 0, 0}, n_sets = 0;
    // bucket path: rows that are (0,0) in every table a sort serves are left out of it (k16_msm_set_zero_row_mask).  d_skip_ac:
    // zero in A and C (A's sort serves both); d_skip_b: zero in B1 and B2, which get a sort of their own (B2's lane) when
    // that drops enough entries to pay for it -- otherwise all four share A's sort and d_skip_ac holds the rows zero in all
    void* d_skip_ac = nullptr;
    void* d_skip_b  = nullptr;
    bool  b_sort    = false;
    bool  b_derive  = false; // B1 / B2 accumulate bucket lists of their own, derived from A's partition without their (0,0) rows
};

// Host side of the compact upload: the context's host threads (k16_ctx_pool) each scan a contiguous range of the witness.
struct WitnessPacker {
    k16_host_pool*        pool      = nullptr;
    unsigned              n_threads = 0; // ranges (= pool width)
    uint32_t              n_vars    = 0;
    uint8_t*              h_narrow  = nullptr; // pinned, device-mapped: one byte per wire
    uint32_t*             h_idx     = nullptr; // pinned: n_threads regions of `cap` wide-value indices
    uint8_t*              h_val     = nullptr; //         ... and their 32-byte values
    uint8_t*              d_narrow  = nullptr;
    uint32_t*             d_idx     = nullptr;
    uint8_t*              d_val     = nullptr;
    uint32_t*             h_bad     = nullptr; // pinned: k_wtns_expand_wide's report of a bad list entry (see WideLists)
    uint32_t*             d_bad     = nullptr;
    uint32_t              cap       = 0; // wide values a range's region holds (a quarter of the range: beyond that, plain copy)
    std::vector<uint32_t> count;
    std::vector<uint8_t>  overflow;

    void pack_range(const uint8_t* src, unsigned t)
    {
        const uint64_t lo = (uint64_t)n_vars * t / n_threads, hi = (uint64_t)n_vars * (t + 1) / n_threads;
        uint32_t*      ix = h_idx + (size_t)t * cap;
        uint8_t*       vv = h_val + (size_t)t * cap * 32;
        uint32_t       c = 0;
        bool           ovf = false;
        for (uint64_t i = lo; i < hi; i++) {
            uint64_t w[4];
            memcpy(w, src + i * 32, 32);
            if (((w[0] >> 8) | w[1] | w[2] | w[3]) == 0) {
                h_narrow[i] = (uint8_t)w[0];
            } else {
                h_narrow[i] = 0;
                if (c < cap) {
                    ix[c] = (uint32_t)i;
                    memcpy(vv + (size_t)c * 32, w, 32);
                    c++;
                } else {
                    ovf = true;
                }
            }
        }
        count[t]    = c;
        overflow[t] = ovf;
    }
    // returns false when a range had more wide values than its region holds (the caller then copies the witness plainly)
    bool pack(const void* witness)
    {
        const uint8_t* src = (const uint8_t*)witness;
        pool->run(n_threads, [&](unsigned t) { pack_range(src, t); });
        for (unsigned t = 0; t < n_threads; t++)
            if (overflow[t]) return false;
        return true;
    }
    uint64_t wide_total() const
    {
        uint64_t s = 0;
        for (unsigned t = 0; t < n_threads; t++) s += count[t];
        return s;
    }
    ~WitnessPacker()
    {
        if (h_narrow) (void)hipHostFree(h_narrow);
        if (h_idx) (void)hipHostFree(h_idx);
        if (h_val) (void)hipHostFree(h_val);
        if (h_bad) (void)hipHostFree(h_bad);
    }
};
static WitnessPacker* packer_create(k16_ctx* ctx, uint32_t n_vars)
{
    k16_host_pool* pool = k16_ctx_pool(ctx);
    if (!pool || n_vars < (1u << 16)) return nullptr; // small circuits: the plain copy is a few microseconds
    // 32 ranges whatever the pool's width (the expansion kernel takes up to 32 wide-value lists): with one range per thread a
    // worker that the host's scheduler parks in the middle of its range holds the whole proof up (observed on a shared
    // 256-CPU host: 1.4-6 ms instead of 0.25 for one proof in four); smaller ranges are handed to whoever is running
    const unsigned T = 32u;
    WitnessPacker* w = new WitnessPacker();
    w->pool          = pool;
    w->n_threads     = T;
    w->n_vars        = n_vars;
    w->cap           = (n_vars / T) / 4 + 64;
    w->count.assign(T, 0);
    w->overflow.assign(T, 0);
    const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
    if (hipHostMalloc((void**)&w->h_narrow, ((size_t)n_vars + 3) & ~(size_t)3, flags) != hipSuccess ||
        hipHostMalloc((void**)&w->h_idx, (size_t)T * w->cap * 4, flags) != hipSuccess ||
        hipHostMalloc((void**)&w->h_val, (size_t)T * w->cap * 32, flags) != hipSuccess ||
        hipHostMalloc((void**)&w->h_bad, 64, flags) != hipSuccess ||
        hipHostGetDevicePointer((void**)&w->d_bad, w->h_bad, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&w->d_narrow, w->h_narrow, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&w->d_idx, w->h_idx, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&w->d_val, w->h_val, 0) != hipSuccess) {
        (void)hipGetLastError();
        delete w;
        return nullptr;
    }
    return w;
}

// number of witness values >= 256 (plain-copy path: small circuits, or a witness too wide for the compact upload)
static uint64_t count_wide_host(k16_ctx* ctx, const void* h_wtns, uint64_t n)
{
    const uint8_t* src = (const uint8_t*)h_wtns;
    auto range = [&](uint64_t lo, uint64_t hi) {
        uint64_t c = 0;
        for (uint64_t i = lo; i < hi; i++) {
            uint64_t w[4];
            memcpy(w, src + i * 32, 32);
            c += ((w[0] >> 8) | w[1] | w[2] | w[3]) != 0;
        }
        return c;
    };
    k16_host_pool* pool = n >= (1u << 16) ? k16_ctx_pool(ctx) : nullptr;
    if (!pool) return range(0, n);
    const unsigned        T = pool->width();
    std::vector<uint64_t> part(T, 0);
    pool->run(T, [&](unsigned t) { part[t] = range(n * t / T, n * (t + 1) / T); });
    uint64_t c = 0;
    for (uint64_t v : part) c += v;
    return c;
}

static std::vector<void*> prover_key_buffers(const k16_prover* p)
{
    return {p->d_A, p->d_B1, p->d_C, p->d_H, p->d_Htab, p->d_B2, p->d_slices, p->d_longs, p->d_rowof, p->d_wire, p->d_coef,
            p->d_shift9, p->d_zmask[0], p->d_zmask[1], p->d_zmask[2], p->d_zmask[3], p->d_skip_ac, p->d_skip_b};
}

static void prover_free(k16_prover* p)
{
    if (!p) return;
    void* own[] = {p->d_wtns, p->d_a, p->d_b, p->d_c, p->d_t[0], p->d_t[1], p->d_t[2], p->d_n16}; // per-proof buffers
    for (void* b : own)
        if (b) (void)hipFree(b);
    if (!p->key) { // a half-built prover: the key part is still its own
        for (void* b : prover_key_buffers(p))
            if (b) (void)hipFree(b);
    }
    p->key.reset(); // (the last prover of a shared key frees it here)
    if (p->st2) (void)hipStreamDestroy(p->st2);
    if (p->ev_w) (void)hipEventDestroy(p->ev_w);
    if (p->ev_h) (void)hipEventDestroy(p->ev_h);
    if (p->cls) k16_scalar_classes_destroy(p->cls);
    delete p->packer;
    delete p;
}

#define K16_HIP_P(ctx, call, p)                                                   \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);       \
            prover_free(p);                                                       \
            return K16_ERR_HIP;                                                   \
        }                                                                         \
    } while (0)

static int prover_finish_create(k16_prover* p, k16_prover** out);

extern "C" int k16_prover_create_mem(k16_ctx* ctx, const void* zkey_bytes, size_t zkey_size, k16_prover** out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !zkey_bytes || !out) return K16_ERR_ARG;
    *out = nullptr;
    BinView bv;
    int     rc = parse_binfile((const uint8_t*)zkey_bytes, zkey_size, "zkey", 1, &bv); // fullprover.cpp:150
    if (rc) {
        ctx->err = "zkey: not an iden3 zkey container (type/version/sections)";
        return rc;
    }
    for (int s = 1; s <= 9; s++) {
        if (s == 3) continue;
        if (!bv.sec[s].p) {
            ctx->err = "zkey: missing section";
            return K16_ERR_FORMAT;
        }
    }
    // zkey_utils.hpp:54-84
    uint32_t proto = 0;
    if (bv.sec[1].size < 4) return K16_ERR_FORMAT;
    memcpy(&proto, bv.sec[1].p, 4);
    if (proto != 1) {
        ctx->err = "zkey file is not groth16";
        return K16_ERR_CURVE; // the reference throws invalid_argument -> UNSUPPORTED_ZKEY_CURVE
    }
    const uint8_t* h    = bv.sec[2].p;
    uint64_t       hsz  = bv.sec[2].size;
    uint32_t       n8q = 0, n8r = 0;
    if (hsz < 4) return K16_ERR_FORMAT;
    memcpy(&n8q, h, 4);
    if (n8q != 32 || hsz < 4 + 32 + 4) {
        ctx->err = "zkey curve not supported";
        return K16_ERR_CURVE;
    }
    memcpy(&n8r, h + 4 + n8q, 4);
    if (n8r != 32 || hsz < 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128 ||
        memcmp(h + 4 + n8q + 4, BN254_R_LE, 32) != 0) { // fullprover.cpp:154-158
        ctx->err = "zkey curve not supported";
        return K16_ERR_CURVE;
    }
    h += 4 + n8q + 4 + n8r;
    k16_prover* p = new k16_prover();
    p->ctx        = ctx;
    // the std containers below (SpMV plan, staging vectors) may throw: the half-built prover then goes with the unwinding
    // (the explicit error paths free it themselves and return normally)
    struct FreeOnUnwind {
        k16_prover* p;
        int         base = std::uncaught_exceptions();
        ~FreeOnUnwind()
        {
            if (std::uncaught_exceptions() > base) prover_free(p);
        }
    } free_on_unwind{p};
    memcpy(&p->n_vars, h, 4);
    memcpy(&p->n_public, h + 4, 4);
    memcpy(&p->domain_size, h + 8, 4);
    h += 12;
    memcpy(&p->alpha1, h, 64);
    h += 64;
    memcpy(&p->beta1, h, 64);
    h += 64;
    memcpy(&p->beta2, h, 128);
    h += 128 + 128; // gamma2 unused by the prover
    memcpy(&p->delta1, h, 64);
    h += 64;
    memcpy(&p->delta2, h, 128);
    p->n_coefs = bv.sec[4].size / 44; // zkey_utils.hpp:84 (integer division absorbs the 4-byte count)
    const uint32_t N = p->domain_size;
    if (N == 0 || (N & (N - 1)) || p->n_vars == 0 || p->n_public + 1 > p->n_vars) {
        ctx->err = "zkey: bad header sizes";
        delete p;
        return K16_ERR_FORMAT;
    }
    while ((1u << p->logN) < N) p->logN++;
    if (bv.sec[5].size < (uint64_t)p->n_vars * 64 || bv.sec[6].size < (uint64_t)p->n_vars * 64 ||
        bv.sec[7].size < (uint64_t)p->n_vars * 128 ||
        bv.sec[8].size < (uint64_t)(p->n_vars - p->n_public - 1) * 64 || bv.sec[9].size < (uint64_t)N * 64 ||
        bv.sec[4].size < 4 + p->n_coefs * 44) {
        ctx->err = "zkey: section shorter than the header implies";
        delete p;
        return K16_ERR_FORMAT;
    }

    // regroup the coefficients into rows (spmv_plan.h): length-sorted 64-row slices + long rows
    const uint8_t* cf = bv.sec[4].p + 4;
    SpmvPlan       plan;
    rc = spmv_plan_build(cf, p->n_coefs, N, p->n_vars, &plan);
    if (rc) {
        ctx->err = rc == -1 ? "zkey: coefficient index out of range" : "zkey: too many coefficients for 32-bit entry offsets";
        delete p;
        return K16_ERR_FORMAT;
    }
    std::vector<SpmvSlice>& slices = plan.slices;
    std::vector<SpmvLong>&  longs  = plan.longs;
    std::vector<uint32_t>&  row_of = plan.row_of;
    std::vector<uint32_t>   wire(std::max<uint64_t>(plan.n_entries, 1), 0);
    std::vector<uint8_t>    vals(std::max<uint64_t>(plan.n_entries, 1) * 32, 0); // padding: coefficient 0 (times wire 0)
    for (uint64_t i = 0; i < p->n_coefs; i++) {
        const size_t pos = plan.pos_of[i];
        uint32_t     sw;
        memcpy(&sw, cf + i * 44 + 8, 4);
        wire[pos] = sw;
        // stored value = coef * 2^512 mod r (canonical); the Fr9 kernels want coef * 2^522: ten modular doublings
        Fr cv;
        memcpy(cv.v, cf + i * 44 + 12, 32);
        for (int d = 0; d < 10; d++) cv = fdbl(cv);
        memcpy(&vals[pos * 32], cv.v, 32);
    }
    p->n_slices = plan.n_slices;
    p->n_long   = plan.n_long;

    K16_HIP_P(ctx, hipSetDevice(ctx->device), p);
    const size_t nv = p->n_vars, nc = p->n_vars - p->n_public - 1;
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_A, std::max<size_t>(nv * 64, 64)), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_B1, std::max<size_t>(nv * 64, 64)), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_B2, std::max<size_t>(nv * 128, 128)), p);
    // the C table is stored with n_public + 1 leading (0,0) rows, i.e. indexed by WIRE like A / B1 / B2: the C MSM
    // (groth16.cpp:106-112: points C[0..), scalars wtns + n_public + 1) then runs over the whole witness and shares the A
    // MSM's bucket sort instead of sorting the same scalars, shifted by two, once more; (0,0) rows contribute nothing
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_C, std::max<size_t>(nv * 64, 64)), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_H, (size_t)N * 64), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_slices, slices.size() * sizeof(SpmvSlice)), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_longs, longs.size() * sizeof(SpmvLong)), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_rowof, row_of.size() * 4), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_wire, wire.size() * 4), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_coef, vals.size()), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_wtns, nv * 32), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_n16, nv * 2 + 64), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_a, (size_t)N * 32), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_b, (size_t)N * 32), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_c, (size_t)N * 32), p);
    for (int k = 0; k < 3; k++) K16_HIP_P(ctx, hipMalloc((void**)&p->d_t[k], (size_t)N * 32), p);
    hipStream_t st = ctx->stream;
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_A, bv.sec[5].p, nv * 64, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_B1, bv.sec[6].p, nv * 64, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_B2, bv.sec[7].p, nv * 128, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemsetAsync(p->d_C, 0, (nv - nc) * 64, st), p);
    if (nc) K16_HIP_P(ctx, hipMemcpyAsync(p->d_C + (nv - nc), bv.sec[8].p, nc * 64, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_H, bv.sec[9].p, (size_t)N * 64, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_slices, slices.data(), slices.size() * sizeof(SpmvSlice), hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_longs, longs.data(), longs.size() * sizeof(SpmvLong), hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_rowof, row_of.data(), row_of.size() * 4, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_wire, wire.data(), wire.size() * 4, hipMemcpyHostToDevice, st), p);
    K16_HIP_P(ctx, hipMemcpyAsync(p->d_coef, vals.data(), vals.size(), hipMemcpyHostToDevice, st), p);
    // The H MSM has uniform 254-bit scalars and is the longest item of a proof: its static table gets precomputed window
    // tables (one bucket set for all digit positions, c = 20: 13 instead of 16 additions per scalar; 1.7 GB at N = 2^21)
    {
        unsigned fc   = 0;
        uint64_t rows = 0;
        k16_msm_fixed_base_info(N, &fc, &rows);
        if (fc && !ctx->tune.no_fixed_base) {
            K16_HIP_P(ctx, hipMalloc((void**)&p->d_Htab, (size_t)rows * 64), p);
            if ((rc = k16_msm_fixed_base_prepare(ctx, K16_G1, p->d_H, N, p->d_Htab))) {
                prover_free(p);
                return rc;
            }
        }
    }
    // G1 tables -> the accumulate kernel's row layout, once (in place; see k16_msm_bases_prepare)
    if ((rc = k16_msm_bases_prepare(ctx, K16_G1, p->d_A, nv, p->d_A)) ||
        (rc = k16_msm_bases_prepare(ctx, K16_G1, p->d_B1, nv, p->d_B1)) ||
        (rc = k16_msm_bases_prepare(ctx, K16_G2, p->d_B2, nv, p->d_B2)) ||
        (rc = k16_msm_bases_prepare(ctx, K16_G1, p->d_C, nv, p->d_C)) ||
        (rc = k16_msm_bases_prepare(ctx, K16_G1, p->d_H, N, p->d_H))) {
        prover_free(p);
        return rc;
    }
    K16_HIP_P(ctx, hipStreamSynchronize(st), p);
    // FFT table for 2 * domainSize (groth16.hpp:96)
    rc = k16_ntt_get_table(ctx, 2ull * N, &p->ntt);
    if (!rc) rc = k16_ntt_build_coset_shift(ctx, p->ntt, N, &p->d_shift9, st);
    if (rc) {
        prover_free(p);
        return rc;
    }
    K16_HIP_P(ctx, hipStreamSynchronize(st), p);
    {
        // the polynomial chain gates the H MSM, the longest item of a proof: its stream gets the highest priority so
        // that its kernels are not queued behind the witness MSMs that run beside it
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (ctx->tune.no_stream_priority) greatest = 0;
        K16_HIP_P(ctx, hipStreamCreateWithPriority(&p->st2, hipStreamNonBlocking, greatest), p);
    }
    K16_HIP_P(ctx, hipEventCreateWithFlags(&p->ev_w, hipEventDisableTiming), p);
    K16_HIP_P(ctx, hipEventCreateWithFlags(&p->ev_h, hipEventDisableTiming), p);
    // One discarded proof of the trivial assignment (wire 0 = 1, everything else 0): the MSM lanes allocate their
    // workspaces, the lane streams and the code objects of every kernel come into being here instead of inside the first
    // request (35 ms instead of 8 through the facade).  Its outcome does not decide anything: a device that cannot prove
    // says so on the first real request.  Not under fault injection, whose counter counts requests.
    p->packer = packer_create(ctx, p->n_vars);
    // the (0,0) rows of the four witness tables (wires a constraint side never mentions): left out of the bucket sorts, and
    // of the scalar classes' lists
    {
        const size_t         mw = ((size_t)p->n_vars + 63) / 64, mb = mw * 8;
        const void*          tab[4] = {p->d_A, p->d_B1, p->d_B2, p->d_C};
        const int            grp[4] = {K16_G1, K16_G1, K16_G2, K16_G1};
        std::vector<uint64_t> hm[4];
        for (int t = 0; t < 4; t++) {
            K16_HIP_P(ctx, hipMalloc(&p->d_zmask[t], mb), p);
            if ((rc = k16_msm_zero_row_mask(ctx, grp[t], tab[t], p->n_vars, p->d_zmask[t]))) {
                prover_free(p);
                return rc;
            }
            hm[t].resize(mw);
            K16_HIP_P(ctx, hipMemcpyAsync(hm[t].data(), p->d_zmask[t], mb, hipMemcpyDeviceToHost, st), p);
        }
        K16_HIP_P(ctx, hipStreamSynchronize(st), p);
        std::vector<uint64_t> ac(mw), bb(mw);
        uint64_t              n_b = 0;
        for (size_t k = 0; k < mw; k++) {
            ac[k] = hm[0][k] & hm[3][k];
            bb[k] = hm[1][k] & hm[2][k];
            n_b += (uint64_t)__builtin_popcountll(bb[k] & ~ac[k]);
        }
        // A second sort (same scalars, B's rows left out; K16_B_SORT=1) pays in latency, not in throughput: measured on the
        // synthetic Keyless-shape key (half of B1 / B2 (0,0)), three alternating runs of 60 proofs on one box: p50 5.88-6.00
        // against 6.00-6.02 ms, but four provers sharing the GPU 161-163 against 165-168 proofs/s (the sort is memory
        // traffic on top of what the other provers' sorts already move; the additions it saves were issue slots nobody was
        // short of there) -- profiles/r04/ab_b_sort.log.  Off by default.
        p->b_sort = ctx->tune.b_sort && p->n_vars >= (1u << 17) && n_b >= p->n_vars / 8;
        if (!p->b_sort)
            for (size_t k = 0; k < mw; k++) ac[k] &= bb[k];
        // K16_B_DERIVE=1: the one partition (A's) serves every table, and B1 / B2 -- when an eighth or more of the wires are
        // (0,0) in both -- get bucket lists of their own from it without those rows (k16_msm_sort_from_lane(derive)): pass 3
        // again (0.2 ms of small kernels on B2's lane), no second partition of the scalars.  Measured against the default
        // (they read A's lists and step over their (0,0) rows in the accumulation, k_accumulate_skip): B2's accumulation
        // 1.06 -> 0.65 ms, p50 the same (5.78-5.98 both), four provers 167 against 171-179 proofs/s -- as with K16_B_SORT the
        // additions saved were not what the proof waits for, and the extra launches cost the other provers; off.
        p->b_derive = !p->b_sort && ctx->tune.b_derive && p->n_vars >= (1u << 17) &&
                      n_b >= p->n_vars / 8;
        K16_HIP_P(ctx, hipMalloc(&p->d_skip_ac, mb), p);
        K16_HIP_P(ctx, hipMalloc(&p->d_skip_b, mb), p);
        K16_HIP_P(ctx, hipMemcpyAsync(p->d_skip_ac, ac.data(), mb, hipMemcpyHostToDevice, st), p);
        K16_HIP_P(ctx, hipMemcpyAsync(p->d_skip_b, bb.data(), mb, hipMemcpyHostToDevice, st), p);
        K16_HIP_P(ctx, hipStreamSynchronize(st), p);
        if (ctx->tune.no_skip_zero_rows) {
            (void)hipFree(p->d_skip_ac);
            (void)hipFree(p->d_skip_b);
            p->d_skip_ac = p->d_skip_b = nullptr;
            p->b_sort    = false;
            p->b_derive  = false;
        }
        // scalar classes (msm_classes.hip): tables with equal masks share a list set
        // (Off by default, K16_CLASSES=1: measured on the synthetic Keyless-shape key the two paths execute the same number of
        // VALU instructions per proof -- 2.07 against 2.19 G wave-instructions, the additions of a witness MSM being one per
        // non-zero digit either way -- and the bucket path, whose kernels start before the NTT passes fill the chip, ends its
        // G2 MSM earlier: p50 5.95-6.03 against 6.06-6.08 ms, four provers 173-177 against 161-165 proofs/s, same box,
        // profiles/r04/ab_witness_classes.log.)
        if (ctx->tune.classes) {
            for (int t = 0; t < 4; t++) {
                int sidx = -1;
                for (int u = 0; u < t && sidx < 0; u++)
                    if (hm[u] == hm[t]) sidx = p->set_of[u];
                if (sidx < 0) {
                    sidx              = p->n_sets++;
                    p->set_mask[sidx] = p->d_zmask[t];
                }
                p->set_of[t] = sidx;
            }
            if ((rc = k16_scalar_classes_create(ctx, p->n_vars, p->n_sets, &p->cls))) {
                prover_free(p);
                return rc;
            }
        }
    }
    // the key part is complete: from here on it is owned by reference count (k16_prover_create_shared hands it to siblings)
    {
        auto owner    = std::make_shared<ProverKeyOwner>();
        owner->device = ctx->device;
        owner->bufs   = prover_key_buffers(p);
        p->key        = std::move(owner);
    }
    return prover_finish_create(p, out);
    });
}

// what k16_prover_create_mem and k16_prover_create_shared have in common once the key part stands: the discarded warm-up proof
static int prover_finish_create(k16_prover* p, k16_prover** out)
{
    k16_ctx* ctx = p->ctx;
#ifdef K16_TESTING
    const bool fault_env = getenv("K16_FAULT_INJECT") != nullptr;
#else
    const bool fault_env = false;
#endif
    if (!ctx->tune.no_warmup && !fault_env) {
        std::vector<uint8_t> w((size_t)p->n_vars * 32, 0);
        w[0] = 1;
        uint8_t one[32] = {1};
        char    js[2048];
        const int wrc = k16_prover_prove_mem(p, w.data(), p->n_vars, one, one, js, sizeof js, nullptr);
        if (wrc < 0) { // not fatal for create (the key is loaded), but never silent: the first request will fail the same way
            fprintf(stderr, "k16_prover_create: warm-up proof failed (%d): %s\n", wrc, ctx->err.c_str());
            p->warmup_rc = wrc;
        }
    }
    *out = p;
    return K16_OK;
}

// A further prover of the SAME key on the SAME device (include/k16.h): shares `other`'s read-only device data by reference
// count and builds only what a proof writes -- witness / polynomial buffers, MSM lanes (the context's), chain stream, upload
// buffers, scalar-class workspace -- plus this context's root table (made on the device in a few ms, not uploaded).
extern "C" int k16_prover_create_shared(k16_ctx* ctx, const k16_prover* other, k16_prover** out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !other || !out) return K16_ERR_ARG;
    *out = nullptr;
    if (!other->key || other->ctx->device != ctx->device) {
        ctx->err = "k16_prover_create_shared: the other prover lives on another device";
        return K16_ERR_ARG;
    }
    if (ctx == other->ctx) {
        ctx->err = "k16_prover_create_shared: a prover needs a context of its own (MSM lanes, staging slots)";
        return K16_ERR_ARG;
    }
    k16_prover* p = new k16_prover();
    struct FreeOnUnwind {
        k16_prover* p;
        int         base = std::uncaught_exceptions();
        ~FreeOnUnwind()
        {
            if (std::uncaught_exceptions() > base) prover_free(p);
        }
    } free_on_unwind{p};
    *p          = *other;      // header values, key pointers, plan sizes, sort choices, the key's owner (+1)
    p->ctx      = ctx;
    p->d_wtns = p->d_a = p->d_b = p->d_c = nullptr; // everything a proof writes is this prover's own
    p->d_t[0] = p->d_t[1] = p->d_t[2] = nullptr;
    p->d_n16  = nullptr;
    p->ntt    = nullptr;
    p->st2    = nullptr;
    p->ev_w = p->ev_h = nullptr;
    p->packer = nullptr;
    p->cls    = nullptr;
    p->last_h.clear();
    p->warmup_rc = 0;
    const size_t   nv = p->n_vars;
    const uint32_t N  = p->domain_size;
    K16_HIP_P(ctx, hipSetDevice(ctx->device), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_wtns, nv * 32), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_n16, nv * 2 + 64), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_a, (size_t)N * 32), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_b, (size_t)N * 32), p);
    K16_HIP_P(ctx, hipMalloc((void**)&p->d_c, (size_t)N * 32), p);
    for (int k = 0; k < 3; k++) K16_HIP_P(ctx, hipMalloc((void**)&p->d_t[k], (size_t)N * 32), p);
    int rc = k16_ntt_get_table(ctx, 2ull * N, &p->ntt);
    if (rc) {
        prover_free(p);
        return rc;
    }
    K16_HIP_P(ctx, hipStreamSynchronize(ctx->stream), p);
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (ctx->tune.no_stream_priority) greatest = 0;
        K16_HIP_P(ctx, hipStreamCreateWithPriority(&p->st2, hipStreamNonBlocking, greatest), p);
    }
    K16_HIP_P(ctx, hipEventCreateWithFlags(&p->ev_w, hipEventDisableTiming), p);
    K16_HIP_P(ctx, hipEventCreateWithFlags(&p->ev_h, hipEventDisableTiming), p);
    p->packer = packer_create(ctx, p->n_vars);
    if (other->cls) { // (K16_CLASSES=1: the lists are per-proof workspace; the masks they read are the key's)
        if ((rc = k16_scalar_classes_create(ctx, p->n_vars, p->n_sets, &p->cls))) {
            prover_free(p);
            return rc;
        }
    }
    return prover_finish_create(p, out);
    });
}

extern "C" int k16_prover_create(k16_ctx* ctx, const char* zkey_path, k16_prover** out)
{
    return k16_guard(ctx, [&]() -> int {
    if (!ctx || !zkey_path || !out) return K16_ERR_ARG;
    *out = nullptr;
    MappedFile mf;
    int        rc = mf.open_ro(zkey_path);
    if (rc) {
        ctx->err = std::string("zkey: cannot open/map ") + zkey_path;
        return rc;
    }
    return k16_prover_create_mem(ctx, mf.base, mf.size, out);
    });
}

extern "C" void k16_prover_destroy(k16_prover* p)
{
    k16_guard_void([&]() {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    prover_free(p);
    });
}

extern "C" int k16_prover_info(const k16_prover* p, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain_size,
                               uint64_t* n_coefs)
{
    return k16_guard((p ? p->ctx : nullptr), [&]() -> int {
    if (!p) return K16_ERR_ARG;
    if (n_vars) *n_vars = p->n_vars;
    if (n_public) *n_public = p->n_public;
    if (domain_size) *domain_size = p->domain_size;
    if (n_coefs) *n_coefs = p->n_coefs;
    return K16_OK;
    });
}

// Fault injection exists only in the TESTING build of the library (libk16_testing.so, -DK16_TESTING; the production
// libk16.so does not read the variable at all, so an inherited environment cannot make a service fail its proofs).
// K16_FAULT_INJECT="hip_after_msm" / "bad_alloc_in_prove" (every prove while it is set) or "<kind>:<k>" (only the k-th
// prove of this process, 1-based): the prove fails with K16_ERR_HIP, or throws std::bad_alloc, after its witness MSMs
// have been enqueued -- the state a device fault or an allocation failure in the middle of a proof leaves behind.
// Test hook for the error paths (tests/test_boundary.py, tests/test_gpu_parity.py).
#ifdef K16_TESTING
static int fault_injected_now()
{
    static std::atomic<long> calls{0};
    const long               k = ++calls;
    const char*              e = getenv("K16_FAULT_INJECT");
    if (!e) return 0;
    int    kind = 0;
    size_t len  = 0;
    if (strncmp(e, "hip_after_msm", 13) == 0) kind = 1, len = 13;
    else if (strncmp(e, "bad_alloc_in_prove", 18) == 0) kind = 2, len = 18;
    if (!kind) return 0;
    return (e[len] != ':' || atol(e + len + 1) == k) ? kind : 0;
}
#else
static inline int fault_injected_now() { return 0; }
#endif

int k16_msm_classified_phase(k16_ctx* ctx, int group, const void* d_prepared, const k16_scalar_classes* cls, int set, int phase);
// prepacked >= 0: the compact hand-off (k16_prover_prove_compact) -- h_wtns is null, the caller has filled the packer's pinned
// buffers and `prepacked` entries of its wide-value list
static int prove_mem_inner(k16_prover* p, const void* h_wtns, uint64_t n_vars, int64_t prepacked, const uint8_t* r_in,
                           const uint8_t* s_in, char* out_json, size_t cap, float* device_ms);

static int prove_guarded(k16_prover* p, const void* h_wtns, uint64_t n_vars, int64_t prepacked, const uint8_t* r_in,
                         const uint8_t* s_in, char* out_json, size_t cap, float* device_ms)
{
    return k16_guard((p ? p->ctx : nullptr), [&]() -> int {
    if (!p || (!h_wtns && prepacked < 0) || !out_json) return K16_ERR_ARG;
    int rc;
    try {
        rc = prove_mem_inner(p, h_wtns, n_vars, prepacked, r_in, s_in, out_json, cap, device_ms);
    } catch (const std::bad_alloc&) {
        rc = K16_ERR_NOMEM;
        try {
            p->ctx->err = "out of host memory during prove";
        } catch (...) {
        }
    } catch (...) {
        rc = K16_ERR_HIP;
    }
    if (rc < 0) {
        // Whatever failed, nothing of this proof may stay behind: MSMs already enqueued are waited for and dropped (the
        // next prove would otherwise pop them as ITS results), the sort-reuse flags and the lane selection are reset,
        // and the chain stream is drained.  The error text of the failure is kept.
        k16_ctx* ctx = p->ctx;
        std::string err;
        try {
            err = ctx->err; // (a second allocation failure here must not skip the clean-up below)
        } catch (...) {
        }
        (void)k16_msm_abort_all(ctx); // overwrites ctx->err only when an MSM in flight failed itself
        if (p->st2) (void)hipStreamSynchronize(p->st2);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); // (the witness expansion: it writes the packer's bad-entry flag)
        ctx->forced_c          = 0;
        ctx->parallel_combine  = false;
        ctx->wait_after_memset = nullptr;
        try {
            if (!err.empty()) ctx->err = err;
        } catch (...) {
        }
    }
    return rc;
    });
}

extern "C" int k16_prover_prove_mem(k16_prover* p, const void* h_wtns, uint64_t n_vars, const uint8_t* r_in,
                                    const uint8_t* s_in, char* out_json, size_t cap, float* device_ms)
{
    if (!h_wtns) return K16_ERR_ARG;
    return prove_guarded(p, h_wtns, n_vars, -1, r_in, s_in, out_json, cap, device_ms);
}

// ---- compact witness hand-off (include/k16.h; SURVEY 8(f).1, the step after prove_mem).  What k16_prover_prove_mem spends its
// first 0.24 ms on -- scanning the 43 MB witness into one byte per wire + the list of the wide values (WitnessPacker) -- is
// work a witness calculator does for free while it computes the wires: it writes the compact form straight into the prover's
// pinned, device-mapped upload buffers and the proof's first kernel starts at once.
extern "C" int k16_prover_compact_buffers(k16_prover* p, uint8_t** narrow, uint32_t** wide_idx, uint8_t** wide_val,
                                          uint64_t* wide_cap)
{
    if (!p || !narrow || !wide_idx || !wide_val || !wide_cap) return K16_ERR_ARG;
    if (!p->packer) {
        try {
            p->ctx->err = "compact hand-off: this prover uploads its witness plainly (fewer than 2^16 wires, or no host pool)";
        } catch (...) {
        }
        return K16_ERR_ARG;
    }
    *narrow   = p->packer->h_narrow;
    *wide_idx = p->packer->h_idx;
    *wide_val = p->packer->h_val;
    *wide_cap = (uint64_t)p->packer->n_threads * p->packer->cap; // (the scan's per-range regions are one contiguous list here)
    return K16_OK;
}

extern "C" int k16_prover_prove_compact(k16_prover* p, uint64_t n_wide, const uint8_t* r_in, const uint8_t* s_in, char* out_json,
                                        size_t cap, float* device_ms)
{
    if (!p || !out_json) return K16_ERR_ARG;
    WitnessPacker* w = p->packer;
    if (!w || n_wide > (uint64_t)w->n_threads * w->cap) {
        try {
            p->ctx->err = w ? "compact hand-off: more wide values than the list holds (use k16_prover_prove_mem)"
                            : "compact hand-off: this prover uploads its witness plainly";
        } catch (...) {
        }
        return K16_ERR_ARG;
    }
    // (the list's entries are checked by the kernel that reads them, k_wtns_expand_wide: a host loop over 27,000 entries and
    // their bytes costs more than the scan this entry point saves)
    return prove_guarded(p, nullptr, p->n_vars, (int64_t)n_wide, r_in, s_in, out_json, cap, device_ms);
}

static int prove_mem_inner(k16_prover* p, const void* h_wtns, uint64_t n_vars, int64_t prepacked, const uint8_t* r_in,
                           const uint8_t* s_in, char* out_json, size_t cap, float* device_ms)
{
    k16_ctx* ctx = p->ctx;
    if (n_vars < p->n_vars) { // the reference does not check; reading past the buffer is not an option here
        ctx->err = "witness has fewer values than the circuit has wires";
        return K16_ERR_FORMAT;
    }
    uint8_t    r_std[32], s_std[32];
    WipeOnExit wipe_r{r_std, 32}, wipe_s{s_std, 32};
    int        rc;
    if (r_in) {
        memcpy(r_std, r_in, 32);
    } else if ((rc = sample_blinding(r_std))) {
        return rc;
    }
    if (s_in) {
        memcpy(s_std, s_in, 32);
    } else if ((rc = sample_blinding(s_std))) {
        return rc;
    }
    if (geq_r(r_std) || geq_r(s_std)) {
        ctx->err = "blinding scalars must be < r";
        return K16_ERR_ARG;
    }

    K16_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t    st = ctx->stream;
    const uint32_t N  = p->domain_size;
    const bool        host_trace = ctx->tune.trace_host;
    const auto        ht0 = std::chrono::steady_clock::now();
    auto              ht  = [&](const char* what) {
        if (host_trace)
            fprintf(stderr, "[k16 host] %-22s %8.1f us\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count());
    };
    K16_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    int64_t n_wide = -1; // wide (>= 256) witness values, counted on the host: the classification then needs no round trip
    // The bad-entry flag belongs to THIS proof's list kernel: cleared before every proof (a failed compact call must not fail
    // the next, plainly uploaded witness -- ADVICE r5) and looked at only when the packed branch ran.  prove_guarded drains
    // the context's stream after a failure, so no kernel of an earlier proof can still write it.
    bool packed_upload = false;
    if (p->packer) *p->packer->h_bad = 0;
    if (p->packer && (prepacked >= 0 || p->packer->pack(h_wtns))) {
        packed_upload    = true;
        WitnessPacker* w = p->packer;
        n_wide           = prepacked >= 0 ? prepacked : (int64_t)w->wide_total();
        WideLists      L;
        L.n_lists = w->n_threads;
        L.n_vars  = p->n_vars;
        L.bad     = w->d_bad;
        uint32_t most = 0;
        for (unsigned t = 0; t < w->n_threads; t++) {
            L.idx[t]   = w->d_idx + (size_t)t * w->cap;
            L.val[t]   = reinterpret_cast<const uint4*>(w->d_val + (size_t)t * w->cap * 32);
            // (compact hand-off: ONE list of `prepacked` entries laid over the regions, which are contiguous)
            L.count[t] = prepacked >= 0 ? (uint32_t)std::min<int64_t>(w->cap, std::max<int64_t>(0, prepacked - (int64_t)t * w->cap))
                                        : w->count[t];
            most       = std::max(most, L.count[t]);
        }
        hipLaunchKernelGGL(k_wtns_expand_narrow, dim3(std::min<uint32_t>((p->n_vars + 255) / 256, 1024u)), dim3(64), 0, st,
                           reinterpret_cast<const uint32_t*>(w->d_narrow), p->d_wtns, p->n_vars, p->d_n16);
        if (most)
            hipLaunchKernelGGL(k_wtns_expand_wide, dim3(std::min<uint32_t>((most + 255) / 256, 64), w->n_threads), dim3(256), 0, st, L, p->d_wtns, p->d_n16);
        K16_HIP(ctx, hipGetLastError());
    } else {
        K16_HIP(ctx, hipMemcpyAsync(p->d_wtns, h_wtns, (size_t)p->n_vars * 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_wtns_n16, dim3((p->n_vars + 255) / 256), dim3(256), 0, st, (const uint4*)p->d_wtns, p->n_vars, p->d_n16);
        if (p->cls) n_wide = (int64_t)count_wide_host(ctx, h_wtns, p->n_vars);
    }
    if (p->cls) {
        ctx->cur_lane = 0;
        if ((rc = k16_scalar_classes_build(ctx, p->cls, p->d_wtns, p->n_vars, p->set_mask, p->n_sets, n_wide))) return rc;
    }

    // The reference overlaps the four witness MSMs with the a/b/c chain through std::async
    // (groth16.cpp:88-112 vs :116-275); here the chain (HBM-bound) runs on a second stream beside the
    // MSMs (integer-issue-bound) and is joined before the H MSM.
    hipStream_t s2 = p->st2;
    ht("witness upload issued");
    K16_HIP(ctx, hipEventRecord(p->ev_w, st));
    K16_HIP(ctx, hipStreamWaitEvent(s2, p->ev_w, 0));
    const unsigned gN = (N + 255) / 256;
    const bool spmv_n16 = !ctx->tune.spmv_full;
    {
        const uint64_t waves = (uint64_t)p->n_slices + p->n_long;
        if (waves)
            hipLaunchKernelGGL(k_spmv, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s2, p->d_slices, p->n_slices, p->d_rowof,
                               p->d_longs, p->n_long, p->d_wire, p->d_coef, p->d_wtns, p->d_a, p->d_b, N, p->logN,
                               spmv_n16 ? (const uint16_t*)p->d_n16 : (const uint16_t*)nullptr);
    }
    hipLaunchKernelGGL(k_mul, dim3(gN), dim3(256), 0, s2, p->d_c, p->d_a, p->d_b, N); // elementwise: same permutation
    // a, b, c together: iNTT passes in place (input already bit-reversed; the last pass stores [tail, coset shift, bit
    // reversal] into d_t[k]), then the forward passes in place on d_t[k] -- six launches for the six transforms
    Fr* src[3] = {p->d_a, p->d_b, p->d_c};
    if ((rc = k16_ntt_coset_chain(ctx, src, p->d_t, 3, N, p->ntt, p->d_shift9, s2))) return rc;
    // the H scalars go to d_a (free since the inverse transform's last pass): written by the H MSM's own counting pass when it
    // is the fixed-base one (ctx->hs_next below), by k_hscalars otherwise
    // (K16_FUSED_HSCALARS=1; measured: the counting pass grows from 40 to 97 us, the 53 us kernel and its gap disappear -- 4 us
    // less between the chain's end and the end of the counting pass, profiles/r04/proof_timeline_fused_h_scalars.txt: the
    // scalars' arithmetic, 46 us of multiply issue, moves with them; off)
    const bool        hs_fuse = ctx->tune.fused_hscalars;
    const bool        hs_in_sort = hs_fuse && p->d_Htab != nullptr;
    if (!hs_in_sort) hipLaunchKernelGGL(k_hscalars, dim3(gN), dim3(256), 0, s2, p->d_a, p->d_t[0], p->d_t[1], p->d_t[2], N);
    K16_HIP(ctx, hipGetLastError());
    K16_HIP(ctx, hipEventRecord(p->ev_h, s2));
    ht("chain enqueued");

    // groth16.cpp:88-112 : the four witness MSMs.  A, B1 and B2 share their scalars (the witness), so the
    // bucket sort of the A MSM is reused for B1 and B2.
    // Witness scalars are mostly bits and bytes: all but the lowest windows are nearly empty, so the bucket
    // reduction (W * 2^c adds) outweighs the accumulation at the reference's c = 16; c = 13 measured best
    // (profiles/r01: 2.65 vs 3.05 ms at n = 2^20).  The H scalars are uniform and keep the automatic choice.
    struct ForcedC {
        k16_ctx* c;
        unsigned saved;
        ForcedC(k16_ctx* cx, unsigned v) : c(cx), saved(cx->forced_c)
        {
            if (!saved && cx_big(cx, v)) c->forced_c = v;
            // witness scalars: one bucket holds ~45 % of the points, short segments keep it parallel (K16_WITNESS_SEG: sweep)
            c->forced_seg = cx->tune.witness_seg ? (unsigned)cx->tune.witness_seg : 32u;
        }
        static bool cx_big(k16_ctx*, unsigned) { return true; }
        ~ForcedC()
        {
            c->forced_c   = saved;
            c->forced_seg = 0;
        }
    };
    G1Xyzz pi_a, pib1, pi_c, pih;
    G2Xyzz pi_b;
    // (lab builds only, -DK16_LAB: K16_LAB_PROBE_NO_WITNESS runs the witness MSMs over ONE point -- WRONG proofs, what the
    // chain + H MSM cost alone; libk16.so has neither the code nor the name)
#ifdef K16_LAB
    static const bool probe_no_witness = getenv("K16_LAB_PROBE_NO_WITNESS") != nullptr;
    const uint64_t    n_wit = probe_no_witness ? 1 : p->n_vars;
#else
    const uint64_t    n_wit = p->n_vars;
#endif
    unsigned wc = 13;
    if (ctx->tune.witness_c) wc = (unsigned)ctx->tune.witness_c;
    if (p->n_vars < (1u << 17)) wc = 0; // small circuits: automatic
    // All five MSMs are enqueued back to back; their host tails (conversion + Horner, ~0.3 ms each, ~1.2 ms for
    // G2) run while later MSMs occupy the GPU.
    // Lane 0: A, B1; lane 2: B2; lane 1: C (A's bucket sort serves all four), then H once the polynomial chain is
    // done.  The fold and weighted-sum stages of one lane leave most CUs idle; the other lanes' kernels fill them.
    struct LaneReset {
        k16_ctx* c;
        ~LaneReset() { c->cur_lane = 0; }
    } lane_reset{ctx};
    hipStream_t s1 = k16_lane_stream(ctx, 1);
    K16_HIP(ctx, hipStreamWaitEvent(s1, p->ev_w, 0)); // witness upload (lane 0's stream)
    if (p->cls) {
        // scalar classes (msm_classes.hip): masked sums for the wires below 256, the ordinary MSM over the few wide ones --
        // whose bucket sort (lane 0's, window size chosen for THEIR number) serves all four tables as before.  B2 first: its
        // G2 arithmetic and its host combine are the longest of the four
        // The masked sums of all four first (they then run beside the SpMV, whose lanes wait on gathers, not beside the NTT
        // passes), the wide parts after, in the same order = the order of the results.
        struct T {
            int         lane, group, tab;
            const void* rows;
        } const order[4] = {{0, K16_G1, 0, p->d_A}, {2, K16_G2, 2, p->d_B2}, {1, K16_G1, 3, p->d_C}, {0, K16_G1, 1, p->d_B1}};
        const bool split = !ctx->tune.no_split_classes;
        for (int phase = split ? 1 : 0; phase <= (split ? 2 : 0); phase++)
            for (int k = 0; k < 4; k++) {
                ctx->cur_lane = order[k].lane;
                if (phase != 1 && k > 0) {
                    ctx->reuse_sort      = true; // lane 0's sort of the wide scalars (A's) serves every table
                    ctx->reuse_sort_lane = 0;
                }
                if ((rc = k16_msm_classified_phase(ctx, order[k].group, order[k].rows, p->cls, p->set_of[order[k].tab], phase))) return rc;
            }
    } else {
        ForcedC fc(ctx, wc);
        const uint64_t* skip_ac = (const uint64_t*)p->d_skip_ac;
        const uint64_t* skip_b  = p->b_sort ? (const uint64_t*)p->d_skip_b : skip_ac;
        // K16_B2_FIRST=1 (lab): B2 leads on lane 2 and owns the SHARED sort, the G1 MSMs read it
        const bool        b2_lead_env = ctx->tune.b2_first;
        const bool        b2_lead = b2_lead_env && !p->b_sort;
        const int         own     = b2_lead ? 2 : 0; // lane whose sort A and C read
        if (p->b_sort || b2_lead) {
            // B2 (G2: the longest of the four) first, on lane 2.  K16_B_SORT: with a bucket sort of its own that leaves out the
            // rows that are (0,0) in B1 and B2 -- half of them in a circuit whose wires mostly sit on one side of a constraint;
            // as sorted entries they would cost a lane of every addition they sit beside.
            K16_HIP(ctx, hipStreamWaitEvent(k16_lane_stream(ctx, 2), p->ev_w, 0));
            ctx->cur_lane  = 2;
            ctx->skip_next = skip_b;
            // (b2_lead: the shared sort contains B's (0,0) rows; B2's accumulation steps over them as it does when it trails)
            if (b2_lead && !ctx->tune.no_acc_skip && !p->b_derive && p->d_skip_ac) ctx->acc_skip_next = (const uint64_t*)p->d_zmask[2];
            if ((rc = k16_msm_enqueue_prepared(ctx, K16_G2, p->d_B2, p->d_wtns, n_wit))) return rc;
        }
        ctx->cur_lane        = 0;
        ctx->reuse_sort      = b2_lead;
        ctx->reuse_sort_lane = own;
        ctx->skip_next       = skip_ac;
        if ((rc = k16_msm_enqueue_prepared(ctx, K16_G1, p->d_A, p->d_wtns, n_wit))) return rc;
        ctx->cur_lane        = 1;
        ctx->reuse_sort      = true; // C is indexed by wire (see k16_prover_create_mem): same scalars, same sort
        ctx->reuse_sort_lane = own;
        ctx->skip_next       = skip_ac;
        if ((rc = k16_msm_enqueue_prepared(ctx, K16_G1, p->d_C, p->d_wtns, n_wit))) return rc;
        // B1 / B2 on a sort that contains their (0,0) rows: the accumulation steps over them (k_accumulate_skip)
        const bool        acc_skip_on = !ctx->tune.no_acc_skip;
        const bool        b_skip = acc_skip_on && !p->b_sort && !p->b_derive && p->d_skip_ac;
        if (p->b_derive && !b2_lead) {
            // B2 first (lane 2): its lists come from lane 0's partition without B's (0,0) rows; B1 (lane 0, after A) reads them
            ctx->cur_lane = 2;
            if ((rc = k16_msm_sort_from_lane(ctx, 0, 1))) return rc;
            ctx->skip_next = (const uint64_t*)p->d_skip_b;
            if ((rc = k16_msm_enqueue_prepared(ctx, K16_G2, p->d_B2, p->d_wtns, n_wit))) return rc;
        }
        const bool b_derived = p->b_derive && !b2_lead;
        // K16_B1_LANE (round 6 experiment): B1 on a lane of its own instead of behind A's MSM on lane 0 -- its accumulation then
        // starts with A's and C's (they all read lane 0's sort) and its tail runs under the chain, not after it
        const int b1_lane = (ctx->tune.b1_lane == 3 && ctx->tune.h_lane != 3) ? 3 : 0;
        if (b1_lane) K16_HIP(ctx, hipStreamWaitEvent(k16_lane_stream(ctx, b1_lane), p->ev_w, 0));
        ctx->cur_lane        = b1_lane;
        ctx->reuse_sort      = true;
        ctx->reuse_sort_lane = (p->b_sort || b_derived) ? 2 : own;
        ctx->skip_next       = b_derived ? (const uint64_t*)p->d_skip_b : skip_b;
        ctx->acc_skip_next   = b_skip ? (const uint64_t*)p->d_zmask[1] : nullptr;
        if ((rc = k16_msm_enqueue_prepared(ctx, K16_G1, p->d_B1, p->d_wtns, n_wit))) return rc;
        if (!p->b_sort && !b2_lead && !b_derived) {
            // B2 (G2: long latency-bound fold / reduction chains) gets lane 2 and reads lane 0's sort, so it runs beside B1
            ctx->cur_lane        = 2;
            ctx->reuse_sort      = true;
            ctx->reuse_sort_lane = 0;
            ctx->skip_next       = skip_ac;
            ctx->acc_skip_next   = b_skip ? (const uint64_t*)p->d_zmask[2] : nullptr;
            if ((rc = k16_msm_enqueue_prepared(ctx, K16_G2, p->d_B2, p->d_wtns, n_wit))) return rc;
        }
    }
    ht("A C B1 B2 enqueued");
    if (const int fault = fault_injected_now()) {
        if (fault == 2) throw std::bad_alloc();
        ctx->err = "injected fault (K16_FAULT_INJECT)";
        return K16_ERR_HIP;
    }
    // groth16.cpp:281-283
    // K16_H_LANE (round 6 experiment): the H MSM on a lane of its own instead of behind C's MSM on lane 1;
    // K16_H_WAIT_FIRST: its wait for the chain issued behind its sort's memset (msm_sort_launch) instead of here
    const int   h_lane = ctx->tune.h_lane;
    hipStream_t sh     = k16_lane_stream(ctx, h_lane);
    ctx->cur_lane      = h_lane;
    if (ctx->tune.h_wait_first && p->d_Htab)
        ctx->wait_after_memset = p->ev_h;
    else
        K16_HIP(ctx, hipStreamWaitEvent(sh, p->ev_h, 0));
    if (p->d_Htab) {
        if (hs_in_sort) {
            ctx->hs_next[0] = p->d_t[0];
            ctx->hs_next[1] = p->d_t[1];
            ctx->hs_next[2] = p->d_t[2];
        }
        if ((rc = k16_msm_enqueue_fixed_base(ctx, K16_G1, p->d_Htab, p->d_a, N))) return rc;
    } else if ((rc = k16_msm_enqueue_prepared(ctx, K16_G1, p->d_H, p->d_a, N))) {
        return rc;
    }
    ctx->cur_lane = 0;
    // groth16.cpp:325-352 : blinding (host; six single scalar multiplications).  Everything that does not
    // need an MSM result is computed now, while the GPU is busy; the rest right after the MSM it needs.
    G1Xyzz d1     = G1Xyzz::from_aff(p->delta1);
    G1Xyzz d1_r   = h_mul(d1, r_std);
    G1Xyzz d1_s   = h_mul(d1, s_std);
    G2Xyzz d2_s   = h_mul(G2Xyzz::from_aff(p->delta2), s_std);
    Fr rr, ss;
    memcpy(rr.v, r_std, 32);
    memcpy(ss.v, s_std, 32);
    Fr      rs = to_mont(fmul(rr, ss)); // = r*s mod r in standard form (groth16.cpp:348-349)
    uint8_t rs_b[32];
    memcpy(rs_b, rs.v, 32);
    WipeOnExit wipe_rr{&rr, sizeof rr}, wipe_ss{&ss, sizeof ss}, wipe_rs{&rs, sizeof rs}, wipe_rsb{rs_b, 32};
    G1Xyzz d1_rs_neg = pneg(h_mul(d1, rs_b));

    ht("H enqueued + host blinding");
    auto finish_b2 = [&]() -> int {
        // B2's combine is 1.2 ms of G2 arithmetic on one thread: on the pool when it is the classified path's (enqueued
        // second, so that it runs under the GPU's H MSM)
        struct Reset {
            k16_ctx* c;
            ~Reset() { c->parallel_combine = false; }
        } reset{ctx};
        ctx->parallel_combine = p->cls != nullptr;
        const int r2          = k16_msm_finish_group(ctx, K16_G2, &pi_b, nullptr);
        if (r2) return r2;
        pi_b = h_madd(pi_b, p->beta2);
        pi_b = h_add(pi_b, d2_s);
        return K16_OK;
    };
    // results come back in enqueue order: classes A, B2, C, B1, H; bucket path B2, A, C, B1, H with a sort of its own for B,
    // A, C, B2, B1, H with derived lists for B (K16_B_DERIVE), A, C, B1, B2, H without
    const bool b2_first = !p->cls && (p->b_sort || ctx->tune.b2_first);
    if (b2_first && (rc = finish_b2())) return rc;
    if ((rc = k16_msm_finish_group(ctx, K16_G1, &pi_a, nullptr))) return rc;
    ht("A finished");
    pi_a          = h_madd(pi_a, p->alpha1);
    pi_a          = h_add(pi_a, d1_r);
    G1Xyzz a_s    = h_mul(pi_a, s_std);
    if (p->cls && (rc = finish_b2())) return rc;
    if ((rc = k16_msm_finish_group(ctx, K16_G1, &pi_c, nullptr))) return rc;
    const bool b2_before_b1 = !p->cls && !b2_first && p->b_derive; // enqueued A, C, B2, B1
    if (b2_before_b1 && (rc = finish_b2())) return rc;
    if ((rc = k16_msm_finish_group(ctx, K16_G1, &pib1, nullptr))) return rc;
    pib1          = h_madd(pib1, p->beta1);
    pib1          = h_add(pib1, d1_s);
    G1Xyzz b1_r   = h_mul(pib1, r_std);
    if (!p->cls && !b2_first && !b2_before_b1 && (rc = finish_b2())) return rc;
    ht("B2 finished");
    // pi_a and pi_b are final: their affine form and decimal strings are made while the GPU still works on the H MSM
    const G1Aff A = to_affine(pi_a);
    const G2Aff B = to_affine(pi_b);
    const std::string js_ab = "{\"pi_a\":[\"" + fq_to_dec(A.x) + "\",\"" + fq_to_dec(A.y) + "\",\"1\"],\"pi_b\":[[\"" +
                              fq_to_dec(B.x.a) + "\",\"" + fq_to_dec(B.x.b) + "\"],[\"" + fq_to_dec(B.y.a) + "\",\"" +
                              fq_to_dec(B.y.b) + "\"],[\"1\",\"0\"]],\"pi_c\":[\"";
    // ... and everything of pi_c but the H term (groth16.cpp:340-352; the order of the group additions does not change the point)
    pi_c = h_add(pi_c, a_s);
    pi_c = h_add(pi_c, b1_r);
    pi_c = h_add(pi_c, d1_rs_neg);
    {
        struct ParallelCombine { // the H MSM's partial sums: the one combine nothing else runs beside (reset on every exit)
            k16_ctx* c;
            explicit ParallelCombine(k16_ctx* cx) : c(cx) { c->parallel_combine = true; }
            ~ParallelCombine() { c->parallel_combine = false; }
        } pc(ctx);
        rc = k16_msm_finish_group(ctx, K16_G1, &pih, nullptr);
    }
    if (rc) return rc;
    ht("H finished");
    K16_HIP(ctx, hipStreamWaitEvent(st, ctx->pend_ev[(ctx->pend_head + k16_ctx::PEND_SLOTS - 1) % k16_ctx::PEND_SLOTS], 0));
    K16_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    K16_HIP(ctx, k16_event_wait(ctx, ctx->ev_b));
    if (device_ms) K16_HIP(ctx, hipEventElapsedTime(device_ms, ctx->ev_a, ctx->ev_b));
    if (packed_upload && *p->packer->h_bad) { // (every MSM of this proof has been consumed: nothing is left behind)
        const uint32_t code = *p->packer->h_bad;
        ctx->err = (code & 1u)   ? "compact witness: wire number out of range in the wide-value list"
                   : (code & 4u) ? "compact witness: a wire is listed twice in the wide-value list"
                                 : "compact witness: a listed wire must have a zero byte in the narrow array";
        return K16_ERR_FORMAT;
    }
    pi_c = h_add(pi_c, pih);

    ht("device joined");
    const G1Aff Cc = to_affine(pi_c);
    // groth16.cpp:378-410 + dump(): keys sorted, no whitespace
    std::string js = js_ab + fq_to_dec(Cc.x) + "\",\"" + fq_to_dec(Cc.y) + "\",\"1\"],\"protocol\":\"groth16\"}";
    if (js.size() + 1 > cap) return K16_ERR_BUFFER;
    memcpy(out_json, js.c_str(), js.size() + 1);
    ht("proof JSON written");
    return (int)js.size();
}

extern "C" int k16_prover_prove_file(k16_prover* p, const char* wtns_path, const uint8_t* r_std, const uint8_t* s_std,
                                     char* out_json, size_t cap, float* device_ms)
{
    return k16_prover_prove_file_timed(p, wtns_path, r_std, s_std, out_json, cap, device_ms, nullptr);
}

// prove_wall_ms: host wall time of the proof itself, i.e. what RS/fullprover.cpp:226-231 brackets (`prover->prove(...)`,
// after the witness file has been opened, mapped and its header checked at :205-221)
extern "C" int k16_prover_prove_file_timed(k16_prover* p, const char* wtns_path, const uint8_t* r_std, const uint8_t* s_std,
                                           char* out_json, size_t cap, float* device_ms, float* prove_wall_ms)
{
    return k16_guard((p ? p->ctx : nullptr), [&]() -> int {
    if (prove_wall_ms) *prove_wall_ms = 0.f;
    if (!p || !wtns_path) return K16_ERR_ARG;
    k16_ctx*   ctx = p->ctx;
    MappedFile mf;
    int        rc = mf.open_ro(wtns_path);
    if (rc) {
        ctx->err = std::string("wtns: cannot open/map ") + wtns_path;
        return rc;
    }
    BinView bv;
    rc = parse_binfile(mf.base, mf.size, "wtns", 2, &bv); // fullprover.cpp:212
    if (rc || !bv.sec[1].p || !bv.sec[2].p || bv.sec[1].size < 4 + 32 + 4) {
        ctx->err = "wtns: malformed container";
        return K16_ERR_FORMAT;
    }
    // wtns_utils.hpp:32-40, fullprover.cpp:216-221
    uint32_t n8 = 0;
    memcpy(&n8, bv.sec[1].p, 4);
    if (n8 != 32 || memcmp(bv.sec[1].p + 4, BN254_R_LE, 32) != 0) {
        ctx->err = "witness uses a different curve than bn128";
        return K16_ERR_CURVE;
    }
    // (the header's nVars is not trusted -- the reference does not even compare it with the key's, SURVEY 8(b): the
    // section's own length says how many values there are, and prove_mem checks that against the circuit)
    uint64_t have = bv.sec[2].size / 32;
    const auto t0 = std::chrono::steady_clock::now();
    rc            = k16_prover_prove_mem(p, bv.sec[2].p, have, r_std, s_std, out_json, cap, device_ms);
    if (prove_wall_ms) *prove_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
    });
}

extern "C" int k16_prover_warmup_status(const k16_prover* p)
{
    return p ? p->warmup_rc : K16_ERR_ARG;
}

extern "C" int k16_prover_last_h(k16_prover* p, void* h_out)
{
    return k16_guard((p ? p->ctx : nullptr), [&]() -> int {
    if (!p || !h_out) return K16_ERR_ARG;
    k16_ctx* ctx = p->ctx;
    K16_HIP(ctx, hipSetDevice(ctx->device));
    // after prove(), d_a holds the H scalars (standard form)
    K16_HIP(ctx, hipMemcpyAsync(h_out, p->d_a, (size_t)p->domain_size * 32, hipMemcpyDeviceToHost, ctx->stream));
    K16_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return K16_OK;
    });
}
