/*
 * k16.h -- C ABI of libk16.so: the MI355X (gfx950) device shim under the Groth16 prover.
 *
 * This is the "thin extern-C shim" between the C++ prover host code and the HIP kernels.
 * Plain pointers and sizes only; every function returns an int status (K16_OK = 0), never
 * throws, and never falls back to a CPU implementation: without a usable HIP device the
 * context cannot be created and every entry point fails with K16_ERR_NO_DEVICE.
 *
 * What each entry point replaces in the reference (paths relative to
 * rust-rapidsnark/rapidsnark/src/ of aptos-labs/keyless-zk-proofs):
 *
 *   k16_msm*            Curve<F>::multiMulByScalar -> ParallelMultiexp::multiexp
 *                       (curve.hpp:209-215, multiexp.cpp:183-245), G1 and G2
 *   k16_ntt             FFT<RawFr>::fft / ::ifft (fft.cpp:192-246)
 *   k16_prover_*        Groth16::makeProver / Prover::prove (groth16.cpp:18-39, 41-360) and the
 *                       FullProverImpl ctor / prove around them (fullprover.cpp:136-250)
 *   k16_field_op_vec,
 *   k16_point_op_vec    RawFq/RawFr raw ops and Curve<F>::add/dbl (fq_raw_generic.cpp:12-233,
 *                       curve.cpp:91-458) exposed as batch kernels for parity tests
 *
 * Data formats are the reference's own: field elements are 32 bytes,
 * little-endian; Fq/Fr values in Montgomery form (R = 2^256) unless stated; G1 affine = x|y
 * (64 B), G2 affine = x.a|x.b|y.a|y.b (128 B), affine infinity = all zero; XYZZ = x|y|zz|zzz
 * (128 B / 256 B), infinity <=> zz == 0; MSM scalars = 32 B standard (non-Montgomery) form.
 */
#ifndef K16_H
#define K16_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    K16_OK             = 0,
    K16_ERR_NO_DEVICE  = -1, /* no HIP device / runtime failure at context creation */
    K16_ERR_HIP        = -2, /* a HIP call or kernel failed; see k16_last_error */
    K16_ERR_ARG        = -3,
    K16_ERR_IO         = -4, /* file open / mmap failure         (-> ZKEY_FILE_LOAD_ERROR) */
    K16_ERR_FORMAT     = -5, /* malformed zkey / wtns            (-> INVALID_INPUT) */
    K16_ERR_CURVE      = -6, /* prime is not BN254 r             (-> UNSUPPORTED_ZKEY_CURVE /
                                                                     WITNESS_GENERATION_INVALID_CURVE) */
    K16_ERR_BUFFER     = -7,
    K16_ERR_NOMEM      = -8  /* a host allocation failed inside the library (-> PROVER_NOT_READY) */
};

enum { K16_G1 = 0, K16_G2 = 1 };
enum { K16_FQ = 0, K16_FR = 1 };
/* Selectors for k16_field_op_vec / k16_point_op_vec only: the SAME inputs and outputs (the reference's canonical
 * Montgomery form), but the operation runs on the representation the hot kernels use -- the unsaturated radix-2^29
 * field of bn254_fq9.h (Fq9 for G1, Fr9 for the NTT chain, Fq2n = Fq2 over Fq9 for G2; elements of K16_FQ2N are 64 B:
 * a | b) and the Eng9 / Eng2n point formulas of the MSM kernels.  Values are converted at the edges, exactly.
 * For these selectors bits 8-11 / 12-15 of `op` add that many multiples of the modulus to operand a / b after the
 * conversion (operands at the documented bounds: X < 8p, Y < 4p, ZZ < 2p ...); the result must not change. */
enum { K16_FQ9 = 2, K16_FR9 = 3, K16_FQ2N = 4,
       K16_FQ2H = 5 /* Fq2 on a LANE PAIR (bn254_fq2pair.h: real part on the even lane, imaginary part on the odd one) -- what
                       the G2 accumulation, fold and weighted-sum kernels execute since round 5; elements as for K16_FQ2N */ };
enum { K16_G1_ENG9 = 2, K16_G2_ENG2N = 3, K16_G2_PAIR = 4 /* the XYZZ formulas instantiated on K16_FQ2H */ };
#define K16_OP_BOUND_A(k) ((k) << 8)
#define K16_OP_BOUND_B(k) ((k) << 12)
enum { K16_OP_ADD = 0, K16_OP_SUB, K16_OP_NEG, K16_OP_MUL, K16_OP_SQR, K16_OP_TOMONT, K16_OP_FROMMONT,
       /* K16_FQ9 / K16_FR9 only: the NTT butterflies' lazy-limb forms (bn254_fq9.h fadd9_lazy, fsub9_lazy4_t) followed by the
        * multiplication that consumes them: (a + b) * b and (a - b) * b; K16_OP_BOUND_A counts in steps of 2 moduli here
        * (up to a + 30 p), b stays below 2 p as in the kernels */
       K16_OP_LAZY_ADDMUL, K16_OP_LAZY_SUBMUL };
enum { K16_PT_ADD = 0, K16_PT_MADD, K16_PT_DBL,
       /* K16_G1_ENG9 only: the bucket accumulation's own mixed addition (bn254_fq9.h acc9_madd: the accumulator keeps
        * W = +-Y with a flag, the entry's sign goes into the addition, lazy subtractions).  p1 XYZZ, p2 affine as for
        * K16_PT_MADD; K16_OP_BOUND_B bit 0: the entry is negative (result p1 - p2), bit 1: the accumulator arrives with
        * W = -Y.  Result: the XYZZ representation of curve.cpp:185-250 for p1 + (+-p2). */
       K16_PT_MADD_ACC };

typedef struct k16_ctx    k16_ctx;
typedef struct k16_prover k16_prover;

/* ---- process set-up (optional) ----
 * ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); one prover uses four
 * streams, so a pool of provers on one GPU, or a prover beside RCCL, wants more.  This sets the
 * variable to n unless the environment already has it.  Call it from the host program's start-up, BEFORE the process's
 * first HIP call and before it starts threads (the library itself never touches the environment). */
int         k16_runtime_hw_queues(int n);
/* number of HIP devices this process can use (0 without a device or runtime; never an error code) */
int         k16_device_count(void);
/* Host worker threads of the library: ONE pool per process, shared by all contexts and provers (the reference has one TBB
 * arena per process, multiexp.cpp:46) -- K16_HOST_THREADS, default 3/4 of the CPUs the process may use, at most 12, counting
 * the calling thread.  Returns the number of workers started so far (0 before the first prover is created). */
int         k16_host_threads(void);

/* ---- context: one per GPU (one process per GPU in multi-GPU runs) ----
 * THREADING: a context (and every prover / key / classes object created from it) is for ONE caller thread at a time.  Two
 * exceptions the library itself relies on: one thread may k16_msm_enqueue* while another k16_msm_finish*es (bench.py), and
 * k16_verify_batch may be called from several threads -- its latency path (n <= 64) shares the context's staging area and is
 * serialised inside the call.  Callers that want proofs in parallel use one context per prover (FullProver's K16_DEVICES
 * pool does); all contexts of a process share one host-thread pool (k16_host_threads). */
int         k16_ctx_create(int device, k16_ctx** out);
/* The same, behind `stream_offset` (0..7) placeholder streams that the context creates first and keeps until it is destroyed.
 * ROCm hands hardware queues to streams in creation order and the queues share four dispatch pipes, so the offset decides WHICH
 * streams of two contexts that work on one GPU take turns on a pipe: with the second prover of a device one stream behind the
 * first (what FullProver does for K16_DEVICES=0,0) two provers reach 198 instead of 189 proofs/s through the facade (8 of 8
 * alternating runs, profiles/r06/ab_facade_pool_stream_offset.log); a context that works ALONE wants offset 0.
 * Results are identical for every offset. */
int         k16_ctx_create_ex(int device, int stream_offset, k16_ctx** out);
void        k16_ctx_destroy(k16_ctx* ctx);
const char* k16_last_error(const k16_ctx* ctx);
/* Tuning.  K16_OPT_PIPELINED_MSM (value 0/1, default 0): the caller keeps several MSMs in flight on different lanes
 * (k16_msm_set_lane) and cares about MSMs per second rather than the latency of one: the weighted bucket sum uses 16
 * instead of 8 slots per lane (12 % less work, chains twice as long) and consecutive bucket accumulations of different
 * lanes are fenced one behind the other (they never overlap anyway; the per-kernel HIP-event statistics then time
 * execution only).  Results are identical either way.
 * K16_OPT_GRAPHS (0/1, default 0): the ~50 launches of an MSM's sort and of its fold + reduction are captured into HIP
 * graphs (per lane, shape and staging slot) after their second use and replayed with one call each -- for hosts whose
 * kernel launches are slow.  The per-stage statistics then cover the bucket accumulation only.
 * K16_OPT_SHARED_GPU (0/1, default 0): other contexts prove on the same GPU at the same time (FullProver sets it for the
 * entries of K16_DEVICES that name a device more than once): kernels leave room for the other provers' (the NTT passes
 * keep three workgroups per CU instead of four) -- more proofs per second, up to 0.3 ms more for a proof alone.  Results
 * are identical either way.
 * K16_OPT_YIELDING_WAITS (0/1, default 0; round 6): the host waits of k16_msm_finish* and of the prove calls poll the device
 * (hipEventQuery) with short sleeps instead of spinning inside the runtime: a caller that waits for its proof costs next to no CPU
 * (a proof otherwise keeps one core busy for its whole 5 ms) at up to ~50 us per wait -- for hosts that run many provers (FullProver
 * sets it for every pool with more than one prover).  Results are identical either way. */
enum { K16_OPT_PIPELINED_MSM = 1, K16_OPT_GRAPHS = 2, K16_OPT_SHARED_GPU = 3, K16_OPT_YIELDING_WAITS = 4 };
int         k16_ctx_set_option(k16_ctx* ctx, int option, int value);
int         k16_sync(k16_ctx* ctx);
/* the HIP stream every kernel of this context is launched on (hipStream_t as void*) */
void*       k16_stream(k16_ctx* ctx);

/* ---- device memory (so callers need no HIP of their own) ---- */
int k16_dev_alloc(k16_ctx* ctx, size_t bytes, void** dptr);
int k16_dev_free(k16_ctx* ctx, void* dptr);
int k16_h2d(k16_ctx* ctx, void* dptr, const void* hptr, size_t bytes);
int k16_d2h(k16_ctx* ctx, void* hptr, const void* dptr, size_t bytes);
/* Page-lock a host buffer the caller owns (hipHostRegister) -- e.g. the scalar array Curve::multiMulByScalar (curve.hpp:209-215)
 * receives by pointer on every call -- so that k16_h2d / k16_msm_host / k16_msm_sharded_run copy from
 * it by DMA instead of through the runtime's staging buffer (a 2^24-row scalar array: 24 instead of 27 ms per upload, and
 * the copy no longer occupies a host core).  The buffer must stay mapped until k16_host_unregister.  Not needed for
 * correctness anywhere. */
int k16_host_register(k16_ctx* ctx, void* hptr, size_t bytes);
int k16_host_unregister(k16_ctx* ctx, void* hptr);

/* ---- timing on the context's stream with HIP events ---- */
int k16_timer_start(k16_ctx* ctx);
int k16_timer_stop(k16_ctx* ctx, float* elapsed_ms); /* synchronises */
/* per-kernel statistics: when enabled, the named hot kernels are bracketed by HIP events.
 * name: "msm_accumulate", "msm_sort", "msm_reduce", "ntt".  total_ms / launches since reset. */
int k16_kernel_stats_enable(k16_ctx* ctx, int on); /* 0 off, 1 all stages, 2 only "msm_accumulate" (+ the host_* timers) */
int k16_kernel_stats_reset(k16_ctx* ctx);
int k16_kernel_stats_get(k16_ctx* ctx, const char* name, uint64_t* launches, double* total_ms);

/* ---- multi-scalar multiplication  (multiexp.cpp:183-245) ----
 * d_bases: n affine points on the device; d_scalars: n x 32 B on the device.
 * Result: XYZZ point (host memory, 128 B for G1 / 256 B for G2), and/or its affine form.
 * Any 256-bit scalar is accepted; (0,0) bases contribute nothing; n == 0 gives infinity.
 * k16_msm / k16_msm_host take any n that fits the device: above 2^24 points they run contiguous chunks of 2^24 on two
 * lanes and fold the chunk results (the in-time version of the multi-GPU sharding).  k16_msm_enqueue* is one device
 * pass and refuses n >= 2^32 / 80. */
int k16_msm(k16_ctx* ctx, int group, const void* d_bases, const void* d_scalars, uint64_t n, void* h_out_xyzz,
            void* h_out_affine);
/* same with host buffers (uploads, runs, downloads) */
int k16_msm_host(k16_ctx* ctx, int group, const void* h_bases, const void* h_scalars, uint64_t n, void* h_out_xyzz,
                 void* h_out_affine);
/* Enqueue only (no host synchronisation): leaves the per-window sums in the context's workspace.
 * k16_msm_finish() synchronises, downloads them and does the final Horner combine on the host. */
int k16_msm_enqueue(k16_ctx* ctx, int group, const void* d_bases, const void* d_scalars, uint64_t n);
int k16_msm_finish(k16_ctx* ctx, void* h_out_xyzz, void* h_out_affine);
/* k16_msm_finish for a caller that knows which group the oldest MSM in flight belongs to: fails with K16_ERR_ARG and
 * consumes nothing when it is the other one (a G2 result is 256 bytes; it must never land in a 128-byte G1 buffer). */
int k16_msm_finish_group(k16_ctx* ctx, int group, void* h_out_xyzz, void* h_out_affine);
/* number of MSMs enqueued and not yet finished (0 .. 8), negative on error */
int k16_msm_pending(k16_ctx* ctx);
/* error recovery: waits for and drops every MSM in flight, clears sort-reuse / lane state (the prover does this on
 * every failed prove so that a later prove never pops a stale result) */
int k16_msm_abort_all(k16_ctx* ctx);
/* Static point tables (the zkey's sections 5-9) can be prepared once: d_out (same size as d_bases)
 * receives the rows in the layout the accumulate kernel gathers from (G1: x*2^261, y*2^261 mod p packed
 * in 2 x 32 B -- the Montgomery form of the kernels' radix-2^29 field; G2: the same per Fq2 component).  A prepared
 * table is passed to k16_msm_enqueue_prepared instead of the zkey-format one. */
int k16_msm_bases_prepare(k16_ctx* ctx, int group, const void* d_bases, uint64_t n, void* d_out);
int k16_msm_enqueue_prepared(k16_ctx* ctx, int group, const void* d_prepared, const void* d_scalars, uint64_t n);
/* Fixed-base MSM for a static G1 table: W = ceil(257/c) precomputed window tables (row w*n + i =
 * 2^(c*w) * P_i) let all digit positions share ONE bucket set, so c can be 20 instead of 16 and a 254-bit scalar costs 13
 * bucket additions instead of 16.  k16_msm_fixed_base_info gives the window size chosen for n (0: not available -- n below
 * 2^13 or n*W >= 2^25 -- use the ordinary MSM) and the table size in 64-byte rows; k16_msm_fixed_base_prepare fills the
 * caller's d_table from a zkey-format table (once; ~W*n*(c + 400) field multiplications); k16_msm_enqueue_fixed_base is
 * k16_msm_enqueue_prepared for such a table (results through k16_msm_finish, same values). */
int k16_msm_fixed_base_info(uint64_t n, unsigned* c, uint64_t* table_rows);
int k16_msm_fixed_base_prepare(k16_ctx* ctx, int group, const void* d_bases, uint64_t n, void* d_table);
int k16_msm_enqueue_fixed_base(k16_ctx* ctx, int group, const void* d_table, const void* d_scalars, uint64_t n);
/* Scalar-class MSM: for scalar vectors that are mostly small -- the wire values of a circom witness: zeros, ones, bytes, a
 * few field elements.  Replaces what the zero-digit skip of the reference's bucket loop (multiexp.cpp:59-65, `if
 * (chunkValue)`) does for its four witness MSMs (groth16.cpp:88-112).  One classification of the scalars serves every
 * table indexed by the same wires:
 *     sum_i s_i P_i  =  sum_{b<8} 2^b (sum of the P_i with s_i < 256 and bit b of s_i set)  +  MSM over the s_i >= 256
 * k16_msm_zero_row_mask:     bit i of d_mask ((n + 63) / 64 x 8 bytes) = row i of the table is (0,0); such rows are left out
 *                            of that table's lists (they add nothing: curve.cpp:185-250).  Works on zkey-format and prepared
 *                            tables alike ((0,0) stays (0,0)).
 * k16_scalar_classes_create: workspace for up to max_n scalars and max_sets (1..4) tables with different zero rows.
 * k16_scalar_classes_build:  classifies d_scalars[0..n) on the current lane's stream; d_zero_masks[s] (may be NULL) is the
 *                            mask of set s.  n_wide_bound < 0: waits and reads the number of scalars >= 256 back; >= 0: the
 *                            caller's upper bound on it (nothing waits; a larger actual count makes k16_msm_finish of every
 *                            MSM enqueued from these classes fail with K16_ERR_ARG instead of returning a wrong sum).
 * k16_msm_enqueue_classified: the MSM of one PREPARED table (k16_msm_bases_prepare) with the classified scalars, on the
 *                            current lane; result through k16_msm_finish*, same value as k16_msm_enqueue_prepared with the
 *                            original scalars.  The classes must not be rebuilt while such an MSM is in flight.
 * k16_scalar_classes_counts: list lengths of the last build, out[s * 8 + b], then the wide count (tests). */
typedef struct k16_scalar_classes k16_scalar_classes;
int  k16_msm_zero_row_mask(k16_ctx* ctx, int group, const void* d_rows, uint64_t n, void* d_mask);
int  k16_scalar_classes_create(k16_ctx* ctx, uint64_t max_n, int max_sets, k16_scalar_classes** out);
void k16_scalar_classes_destroy(k16_scalar_classes* cls);
int  k16_scalar_classes_build(k16_ctx* ctx, k16_scalar_classes* cls, const void* d_scalars, uint64_t n,
                              const void* const* d_zero_masks, int n_sets, int64_t n_wide_bound);
int  k16_scalar_classes_counts(k16_ctx* ctx, const k16_scalar_classes* cls, uint32_t* out);
int  k16_msm_enqueue_classified(k16_ctx* ctx, int group, const void* d_prepared, const k16_scalar_classes* cls, int set);
/* The next k16_msm_enqueue / _prepared leaves the points whose bit is set in d_mask (k16_msm_zero_row_mask of its table, or
 * the AND of the masks of several tables that will share the sort) out of its bucket sort: (0,0) rows add nothing, but as
 * sorted entries they cost a lane of every addition they sit beside.  Covers one enqueue; n <= 2^24. */
int  k16_msm_set_zero_row_mask(k16_ctx* ctx, const void* d_mask);
/* The next k16_msm_enqueue_prepared on the current lane reads the bucket sort that `lane` has made of the SAME scalar
 * array (same device pointer, n and window size; groth16.cpp:88-112: four MSMs over one witness) instead of sorting.
 *   derive = 0  that lane's lists as they are (the zero-row mask named for this enqueue must be the one they were made with);
 *   derive = 1  lists of the current lane's own, built from that lane's partition without the rows of this enqueue's
 *               zero-row mask (a superset of the owner's): for a table with many (0,0) rows of its own.  Other lanes may
 *               share the derived lists with derive = 0.  K16_ERR_ARG when that lane holds no matching sort. */
int  k16_msm_sort_from_lane(k16_ctx* ctx, int lane, int derive);
/* A context has K16_MSM_LANES independent MSM lanes (HIP stream + workspace).  The next k16_msm_enqueue* uses the
 * selected lane; MSMs on different lanes may overlap on the GPU (their inputs must already be complete: uploads
 * through k16_h2d are).  k16_msm_finish still returns results in enqueue order.  Default lane 0. */
#define K16_MSM_LANES 4
int k16_msm_set_lane(k16_ctx* ctx, int lane);
/* override the window size chosen for the next MSMs (0 = automatic) */
int k16_msm_set_window_bits(k16_ctx* ctx, unsigned c);

/* ---- ONE MSM sharded over several GPUs (BASELINE config 5: "MSMs above ~2^24 points shard the scalar / point
 * array across the GPUs of a node") -- the multi-device form of Curve::multiMulByScalar -> ParallelMultiexp::multiexp
 * (curve.hpp:209-215, multiexp.cpp:183-245).  Shard r owns the contiguous rows [lo_r, hi_r) (the first n % shards shards
 * get one more row) of the point table and of the scalars; every shard runs the whole device pipeline on its rows; the only
 * thing that leaves a device is the shard's partial result, ONE XYZZ point (128 / 256 B), and the results are folded with
 * EC additions.
 *
 * (1) One process, several devices: one context per entry of `devices` (entries may repeat: two shards on one GPU is how
 * one-GPU boxes test it).  The point table is placed once (static tables: k16_msm_sharded_set_bases* converts each slice to
 * the kernels' row layout on its device); every run uploads / reads the scalars per shard on one host thread per shard and
 * folds the partials on the host -- they arrive there anyway for the Horner combine, so no collective is needed.
 * k16_msm_sharded_ctx gives the shard's context, e.g. to fill its slice with k16_synth_points. */
typedef struct k16_msm_shards k16_msm_shards;
int         k16_msm_sharded_create(const int* devices, int n_devices, int group, uint64_t n, k16_msm_shards** out);
void        k16_msm_sharded_destroy(k16_msm_shards* s);
int         k16_msm_sharded_count(const k16_msm_shards* s);
int         k16_msm_sharded_range(const k16_msm_shards* s, int shard, uint64_t* lo, uint64_t* hi);
k16_ctx*    k16_msm_sharded_ctx(k16_msm_shards* s, int shard);
const char* k16_msm_sharded_last_error(const k16_msm_shards* s);
/* h_bases: all n rows on the host, the reference's format (affine, Montgomery, LE) */
int         k16_msm_sharded_set_bases(k16_msm_shards* s, const void* h_bases);
/* ... or shard `shard`'s rows [lo, hi) already on that shard's device, same format */
int         k16_msm_sharded_set_bases_device(k16_msm_shards* s, int shard, const void* d_slice);
/* h_scalars: n x 32 B on the host (any 256-bit values).  Result as for k16_msm. */
int         k16_msm_sharded_run(k16_msm_shards* s, const void* h_scalars, void* h_out_xyzz, void* h_out_affine);
/* d_scalars[r]: shard r's (hi_r - lo_r) x 32 B already on shard r's device */
int         k16_msm_sharded_run_device(k16_msm_shards* s, const void* const* d_scalars, void* h_out_xyzz, void* h_out_affine);
/* Rows per device pass inside a shard (same result for any value): scalars from host memory are uploaded and enqueued in
 * pieces of host_rows (default 2^22 = 128 MB: piece i + 1 crosses PCIe while piece i is sorted and accumulated, two MSMs and
 * one upload in flight), resident scalars in passes of device_rows (default and maximum 2^24).  0 keeps / restores the
 * default; K16_ERR_ARG below 64 or above 2^24. */
int         k16_msm_sharded_set_piece_rows(k16_msm_shards* s, uint64_t host_rows, uint64_t device_rows);
/* wall time of the last run: all shards (upload + device work + per-shard combine, in parallel), the fold, the total */
int         k16_msm_sharded_last_ms(const k16_msm_shards* s, double* shards_ms, double* fold_ms, double* total_ms);
/* (2) One PROCESS per GPU (a launcher starts the ranks; each rank has its own context and its shard): the ranks exchange
 * their partial results with ONE ncclAllGather over xGMI and every rank folds them in rank order (RCCL has no elliptic-curve
 * reduction operator).  RCCL is dlopen'ed at the first call -- libk16.so does not link it; K16_ERR_NO_DEVICE (and
 * k16_rank_comm_load_error) when it cannot be loaded.  Rank 0 makes the 128-byte id and hands it to all ranks by whatever
 * the launcher offers (a file, MPI, a TCP store); k16_rank_comm_create is collective (ncclCommInitRank).
 * K16_RCCL_LIB (environment, read once per process at the first k16_rank_comm_* call) names the RCCL library file outright,
 * for installations outside the loader's search path.  A gather is waited for at most K16_RANK_COMM_TIMEOUT_MS (read by
 * k16_rank_comm_create; default 60 000): when a rank does not arrive in time the communicator is aborted, the call and every
 * later one on it return K16_ERR_HIP -- a dead rank never hangs the others. */
typedef struct k16_rank_comm k16_rank_comm;
int         k16_rank_comm_unique_id(void* out128);
const char* k16_rank_comm_load_error(void);
int         k16_rank_comm_create(k16_ctx* ctx, int rank, int world, const void* unique_id128, k16_rank_comm** out);
void        k16_rank_comm_destroy(k16_rank_comm* c);
int         k16_rank_comm_allgather_fold(k16_rank_comm* c, int group, const void* h_partial_xyzz, void* h_out_xyzz,
                                         void* h_out_affine);
/* The same exchange split in two, for a caller that keeps the GPU busy with the next MSM meanwhile: _start enqueues the
 * upload of this rank's partial, the ncclAllGather and the download on the communicator's stream and returns at once (up to
 * 4 exchanges in flight, K16_ERR_ARG beyond); _finish completes the OLDEST one (bounded wait) and folds it.  Every rank must
 * start its exchanges in the same order. */
int         k16_rank_comm_allgather_start(k16_rank_comm* c, int group, const void* h_partial_xyzz);
int         k16_rank_comm_allgather_finish(k16_rank_comm* c, void* h_out_xyzz, void* h_out_affine);

/* combine partial MSM results from several shards/GPUs: out = sum_i parts[i] (XYZZ, host) */
int k16_points_sum(int group, const void* h_parts_xyzz, uint64_t count, void* h_out_xyzz, void* h_out_affine);

/* ---- NTT over Fr (fft.cpp:192-246), in place on the device, Montgomery form ----
 * max_domain selects the root table (the reference builds it for 2*domainSize, groth16.hpp:96);
 * n must be a power of two <= max_domain. inverse != 0 -> FFT::ifft. */
int k16_ntt(k16_ctx* ctx, void* d_a, uint64_t n, uint64_t max_domain, int inverse);
int k16_ntt_host(k16_ctx* ctx, void* h_a, uint64_t n, uint64_t max_domain, int inverse);

/* ---- synthetic inputs: d_out[i] = (start + i + 1) * G, affine Montgomery (the point family of the
 * reference's own MSM test, alt_bn128_test.cpp:183-190); used by bench.py and the full-size tests ---- */
int k16_synth_points(k16_ctx* ctx, int group, uint64_t start, uint64_t n, void* d_out_affine);
/* d_out[i] = scalar_i * G for n arbitrary 256-bit scalars (device, 32 B little-endian standard form): lets a test build a
 * Groth16 key from a known trapdoor, i.e. a synthetic key of any size whose proofs verify (tests/valid_key_builder.py) */
int k16_synth_points_scalars(k16_ctx* ctx, int group, const void* d_scalars, uint64_t n, void* d_out_affine);

/* ---- batch primitives, for parity tests of the device arithmetic ---- */
int k16_field_op_vec(k16_ctx* ctx, int field, int op, const void* h_a, const void* h_b, void* h_r, uint64_t n);
int k16_point_op_vec(k16_ctx* ctx, int group, int op, const void* h_p1, const void* h_p2, void* h_r, uint64_t n);

/* ---- Groth16 prover (groth16.cpp:41-360 behind fullprover.cpp:136-250) ----
 * k16_prover_create parses the zkey (iden3 binfile, sections 1,2,4-9), checks r, and uploads
 * coefficients and point tables to HBM once.  k16_prover_prove_* run the whole proof on the GPU
 * and write the compact snarkjs JSON (same bytes as Proof::toJson().dump()).
 * r_std / s_std: blinding scalars, 32 B standard form, < r.  NULL => drawn from the OS CSPRNG
 * exactly as groth16.cpp:296-316 does (254-bit candidates, rejection). */
int  k16_prover_create(k16_ctx* ctx, const char* zkey_path, k16_prover** out);
int  k16_prover_create_mem(k16_ctx* ctx, const void* zkey_bytes, size_t zkey_size, k16_prover** out);
/* A further prover of the SAME key on the SAME device as `other` (throughput mode: several provers sharing one GPU, what
 * FullProver builds for K16_DEVICES=0,0; replaces the per-instance zkey mapping + section walk of fullprover.cpp:136-181): it shares other's read-only device data -- point tables, H window tables,
 * coefficients, masks, 2.7 GB at the Keyless shape -- by reference count instead of parsing, uploading and preparing the key
 * again (0.1 s instead of 1.1 s), and owns only what a proof writes.  ctx must be a context of its own on that device.  The
 * shared part is freed with the last prover that uses it; `other` may be destroyed first.  Same proofs as a prover made by
 * k16_prover_create. */
int  k16_prover_create_shared(k16_ctx* ctx, const k16_prover* other, k16_prover** out);
void k16_prover_destroy(k16_prover* p);
int  k16_prover_info(const k16_prover* p, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain_size,
                     uint64_t* n_coefs);
int  k16_prover_prove_file(k16_prover* p, const char* wtns_path, const uint8_t* r_std, const uint8_t* s_std,
                           char* out_json, size_t cap, float* device_ms);
/* the same, also reporting the host wall time of the proof proper (witness file already opened, mapped and checked):
 * the interval the reference's `prover_time` metric covers (fullprover.cpp:226-244) */
int  k16_prover_prove_file_timed(k16_prover* p, const char* wtns_path, const uint8_t* r_std, const uint8_t* s_std,
                                 char* out_json, size_t cap, float* device_ms, float* prove_wall_ms);
/* witness already in memory: n_vars x 32 B standard form (the payload of wtns section 2) */
int  k16_prover_prove_mem(k16_prover* p, const void* h_wtns, uint64_t n_vars, const uint8_t* r_std,
                          const uint8_t* s_std, char* out_json, size_t cap, float* device_ms);
/* Compact witness hand-off (round 5; SURVEY 8(f).1, the step after k16_prover_prove_mem; replaces the file hand-off of
 * fullprover.cpp:212-224 for a caller that owns its witness calculator).  k16_prover_prove_mem spends its first 0.24 ms scanning
 * the n_vars x 32-byte witness into the form that crosses PCIe: one byte per wire -- the value when it is below 256, 0
 * otherwise -- plus the list of the wires that hold larger values (wire number, 32-byte standard-form value; any order, every
 * such wire exactly once).  A witness calculator produces that form for free while it computes the wires:
 *   k16_prover_compact_buffers   the prover's pinned, device-mapped upload buffers (valid for the prover's lifetime; written
 *                                by the caller before every k16_prover_prove_compact): narrow[n_vars], wide_idx[*wide_cap],
 *                                wide_val[*wide_cap][32].  K16_ERR_ARG for a prover without them (circuits below 2^16 wires).
 *   k16_prover_prove_compact     proves the witness the buffers hold, n_wide entries of the list being valid: the proof's
 *                                first kernel starts at once.  Same result, outputs and errors as k16_prover_prove_mem for
 *                                the same witness; K16_ERR_ARG when n_wide exceeds the list (such a witness goes through
 *                                k16_prover_prove_mem), K16_ERR_FORMAT for a wire number out of range, a listed wire whose
 *                                narrow byte is not 0, or a wire listed twice (checked on the device by the kernel that reads
 *                                the list: the entry is skipped, the call fails when its device work has been joined).
 * One proof at a time per prover, as for every prove call; k16_prover_prove_mem / _prove_file overwrite the same buffers. */
int  k16_prover_compact_buffers(k16_prover* p, uint8_t** narrow, uint32_t** wide_idx, uint8_t** wide_val, uint64_t* wide_cap);
int  k16_prover_prove_compact(k16_prover* p, uint64_t n_wide, const uint8_t* r_std, const uint8_t* s_std, char* out_json,
                              size_t cap, float* device_ms);
/* The drop-in FullProver (include/k16_fullprover.hpp) with the witness in memory: `fullprover` points to a FullProver object
 * (the one the Rust crate holds; its pool of provers -- K16_DEVICES -- serves concurrent callers), wtns_values is the payload
 * of the .wtns file's section 2.  FullProver::prove(path) maps and parses a 43 MB file inside every call; a service that
 * keeps several GPUs busy binds this instead.  Returns the JSON's length, or K16_ERR_NO_DEVICE (the reference's
 * PROVER_NOT_READY: not constructed, or every device retired), K16_ERR_HIP (device fault: the slot was rebuilt, retry),
 * K16_ERR_FORMAT / K16_ERR_ARG / K16_ERR_BUFFER / K16_ERR_NOMEM.  prover_time_ms: wall time of the call (may be NULL). */
int  k16_fullprover_prove_mem(const void* fullprover, const void* wtns_values, uint64_t n_values, char* out_json, size_t cap,
                              int* prover_time_ms);
/* The compact hand-off (k16_prover_compact_buffers / k16_prover_prove_compact above) THROUGH THE POOL: the buffers belong to
 * one prover, so the caller leases a slot first, writes the witness into that slot's buffers, and proves on it:
 *   k16_fullprover_compact_lease     waits for a free prover like prove() does and hands out its upload buffers (same meaning
 *                                    as k16_prover_compact_buffers) and the circuit's wire count; *lease names the slot.
 *                                    K16_ERR_NO_DEVICE: not ready / every device retired; K16_ERR_ARG: the key's provers upload
 *                                    plainly (below 2^16 wires) -- nothing is leased then.
 *   k16_fullprover_prove_compact     proves the witness the leased buffers hold and gives the slot back WHATEVER the outcome
 *                                    (results and error classes as k16_fullprover_prove_mem; a device fault rebuilds the slot).
 *   k16_fullprover_compact_cancel    gives the slot back without proving (the witness calculator failed).
 * A lease is for one proof and one thread; a service bounds how long it holds one (the slot serves nobody else meanwhile).
 * Replaces the temp-file hand-off of fullprover.cpp:204-250 / prover_handler.rs:511-527 for a pooled multi-GPU service. */
int  k16_fullprover_compact_lease(const void* fullprover, void** lease, uint8_t** narrow, uint32_t** wide_idx,
                                  uint8_t** wide_val, uint64_t* wide_cap, uint32_t* n_vars);
int  k16_fullprover_prove_compact(const void* fullprover, void* lease, uint64_t n_wide, char* out_json, size_t cap,
                                  int* prover_time_ms);
int  k16_fullprover_compact_cancel(const void* fullprover, void* lease);
/* debugging / parity: H scalars of the last proof (domain_size x 32 B, standard form) */
int  k16_prover_last_h(k16_prover* p, void* h_out);
/* status of the discarded warm-up proof k16_prover_create runs (K16_OK, or the error it ended with: a prover whose device
 * cannot prove reports it here instead of on the first real request) */
int  k16_prover_warmup_status(const k16_prover* p);

/* ---- batched Groth16 verification ----
 * Replaces the CPU check the service runs on every proof before it is released
 * (prover-service/src/request_handler/prover_handler.rs:329-336: Groth16Proof::verify_proof(public_inputs_hash, &pvk), i.e.
 * ark-groth16 0.4.0 prepare_inputs + verify_proof_with_prepared_inputs over ark-ec 0.4.2 / ark-bn254 0.4.0 -- third-party
 * crates, Cargo.lock:501-639; types.rs:141-196 prepared_vk builds the key from the decimal strings of the vkey file):
 *     e(A, B) * e(vk_x, -gamma) * e(C, -delta) == e(alpha, beta),     vk_x = IC[0] + sum_i x_i * IC[i + 1]
 * Points are affine, Montgomery form, little-endian -- the zkey's own point format: G1 64 B, G2 128 B.
 * k16_vk_create uploads the key and computes e(alpha, beta) once (PreparedVerifyingKey::alpha_g1_beta_g2).
 * k16_verify_batch: h_proofs = n x 256 B (A | B | C), h_inputs = n x (n_ic - 1) x 32 B standard-form integers (any 256-bit
 * value; they act modulo r, as Fr::from_le_bytes_mod_order), out_ok[i] = 1 accept / 0 reject.  3n Miller loops and n final
 * exponentiations, one GPU lane each: a throughput path for batches (BASELINE config 4 releases 64 proofs per wave).
 * Input validation: a proof whose A / B / C has a coordinate >= p (a non-canonical encoding: A, A + p, A + 2p would
 * otherwise be interchangeable) or does not lie on the curve (A, C) / the twist (B) is REJECTED (flag 0), as ark's
 * deserialisation refuses it before verify_proof runs.  Membership of B in the r-torsion subgroup is NOT tested (the
 * service verifies its own prover's output; aptos-types validates foreign points when it deserialises them): a caller that
 * verifies proofs from elsewhere must do that first.  Points with a zero coordinate pair count as the point at infinity
 * (the pair then contributes 1, as ark-ec's multi_miller_loop skips it). */
typedef struct k16_vk k16_vk;
int  k16_vk_create(k16_ctx* ctx, const void* alpha1_g1, const void* beta2_g2, const void* gamma2_g2, const void* delta2_g2,
                   const void* ic_g1, uint32_t n_ic, k16_vk** out);
void k16_vk_destroy(k16_vk* vk);
int  k16_verify_batch(k16_ctx* ctx, const k16_vk* vk, const void* h_proofs, const void* h_inputs, uint64_t n,
                      uint8_t* h_out_ok);
/* Small batches (n <= K16_VERIFY_COOP_MAX, default 2048) of k16_verify_batch run ONE WAVEFRONT PER PROOF (a static program
 * of ~2000 steps of <= 64 independent field operations, csrc/verify_script.h): 1-2 ms for one proof or for a wave of 64,
 * where one lane per pairing needs ~45 ms -- the latency the per-proof check of prover_handler.rs:329-336 needs.  Same
 * flags.  parity tests: the GT value that path computes for every proof, e(A,B) e(vk_x,-gamma) e(C,-delta), 12 x 32 B per proof */
int  k16_verify_coop_gt(k16_ctx* ctx, const k16_vk* vk, const void* h_proofs, const void* h_inputs, uint64_t n, void* h_out_gt);
/* parity tests: out[i] = e(P_i, Q_i) exactly as ark-ec's Bn::pairing gives it, 12 x 32 B per value (Fq12 = Fq6[w]/(w^2 - v),
 * Fq6 = Fq2[v]/(v^3 - 9 - u): c0.c0.a, c0.c0.b, c0.c1.a, ...), Montgomery form */
int  k16_pairing_vec(k16_ctx* ctx, const void* h_g1, const void* h_g2, uint64_t n, void* h_out_gt);

#ifdef __cplusplus
}
#endif
#endif
