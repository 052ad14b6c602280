"""ctypes binding of libk16.so (include/k16.h) -- used by tests/, bench.py and __graft_entry__.py.

There is deliberately no fallback: if the HIP library is missing or no GPU is present,
loading / context creation raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("K16_LIB_PATH") or os.path.join(HERE, "libk16.so")  # override: A/B-testing builds

G1, G2 = 0, 1
OPT_PIPELINED_MSM = 1
OPT_GRAPHS = 2
OPT_SHARED_GPU = 3
OPT_YIELDING_WAITS = 4
FQ, FR = 0, 1
FQ9, FR9, FQ2N, FQ2H = 2, 3, 4, 5          # the same ops on the hot kernels' radix-2^29 representations (include/k16.h)
G1_ENG9, G2_ENG2N, G2_PAIR = 2, 3, 4
OP_ADD, OP_SUB, OP_NEG, OP_MUL, OP_SQR, OP_TOMONT, OP_FROMMONT, OP_LAZY_ADDMUL, OP_LAZY_SUBMUL = range(9)
PT_ADD, PT_MADD, PT_DBL, PT_MADD_ACC = range(4)
AFF_BYTES = {G1: 64, G2: 128, G1_ENG9: 64, G2_ENG2N: 128, G2_PAIR: 128}
XYZZ_BYTES = {G1: 128, G2: 256, G1_ENG9: 128, G2_ENG2N: 256, G2_PAIR: 256}


def op_bound(op, ka=0, kb=0):
    """K16_OP_BOUND_A / _B: operands moved up by ka / kb multiples of the modulus (radix-2^29 selectors only)."""
    return op | (ka << 8) | (kb << 12)

ERR = {0: "OK", -1: "NO_DEVICE", -2: "HIP", -3: "ARG", -4: "IO", -5: "FORMAT", -6: "CURVE", -7: "BUFFER", -8: "NOMEM"}

# every symbol include/k16.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "k16_runtime_hw_queues", "k16_device_count", "k16_host_threads", "k16_ctx_create", "k16_ctx_create_ex", "k16_ctx_destroy", "k16_last_error", "k16_sync", "k16_stream",
    "k16_dev_alloc", "k16_dev_free", "k16_h2d", "k16_d2h", "k16_host_register", "k16_host_unregister",
    "k16_timer_start", "k16_timer_stop", "k16_kernel_stats_enable", "k16_kernel_stats_reset", "k16_kernel_stats_get",
    "k16_ctx_set_option", "k16_msm", "k16_msm_host", "k16_msm_enqueue", "k16_msm_finish", "k16_msm_finish_group", "k16_msm_pending", "k16_msm_abort_all", "k16_msm_bases_prepare", "k16_msm_enqueue_prepared", "k16_msm_fixed_base_info", "k16_msm_fixed_base_prepare", "k16_msm_enqueue_fixed_base", "k16_msm_set_window_bits", "k16_msm_set_lane", "k16_points_sum",
    "k16_msm_zero_row_mask", "k16_msm_set_zero_row_mask", "k16_msm_sort_from_lane", "k16_scalar_classes_create", "k16_scalar_classes_destroy", "k16_scalar_classes_build", "k16_scalar_classes_counts", "k16_msm_enqueue_classified",
    "k16_ntt", "k16_ntt_host", "k16_synth_points", "k16_synth_points_scalars", "k16_field_op_vec", "k16_point_op_vec",
    "k16_prover_create", "k16_prover_create_mem", "k16_prover_create_shared", "k16_prover_destroy", "k16_prover_info",
    "k16_prover_prove_file", "k16_prover_prove_file_timed", "k16_prover_prove_mem", "k16_prover_compact_buffers", "k16_prover_prove_compact", "k16_fullprover_prove_mem", "k16_fullprover_compact_lease", "k16_fullprover_prove_compact", "k16_fullprover_compact_cancel", "k16_prover_last_h", "k16_prover_warmup_status",
    "k16_vk_create", "k16_vk_destroy", "k16_verify_batch", "k16_verify_coop_gt", "k16_pairing_vec",
    "k16_msm_sharded_create", "k16_msm_sharded_destroy", "k16_msm_sharded_count", "k16_msm_sharded_range", "k16_msm_sharded_ctx",
    "k16_msm_sharded_last_error", "k16_msm_sharded_set_bases", "k16_msm_sharded_set_bases_device", "k16_msm_sharded_run",
    "k16_msm_sharded_run_device", "k16_msm_sharded_set_piece_rows", "k16_msm_sharded_last_ms",
    "k16_rank_comm_unique_id", "k16_rank_comm_load_error", "k16_rank_comm_create", "k16_rank_comm_destroy", "k16_rank_comm_allgather_fold", "k16_rank_comm_allgather_start", "k16_rank_comm_allgather_finish",
]

_lib = None


class K16Error(RuntimeError):
    def __init__(self, rc, msg=""):
        super().__init__("k16 error %s (%d) %s" % (ERR.get(rc, "?"), rc, msg))
        self.rc = rc


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise K16Error(-1, "libk16.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, u32, sz = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_size_t
    L.k16_runtime_hw_queues.argtypes = [i32]
    L.k16_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.k16_ctx_create_ex.argtypes = [i32, i32, C.POINTER(vp)]
    L.k16_ctx_destroy.argtypes = [vp]
    L.k16_ctx_destroy.restype = None
    L.k16_last_error.argtypes = [vp]
    L.k16_last_error.restype = C.c_char_p
    L.k16_sync.argtypes = [vp]
    L.k16_stream.argtypes = [vp]
    L.k16_stream.restype = vp
    L.k16_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.k16_dev_free.argtypes = [vp, vp]
    L.k16_h2d.argtypes = [vp, vp, vp, sz]
    L.k16_d2h.argtypes = [vp, vp, vp, sz]
    L.k16_host_register.argtypes = [vp, vp, sz]
    L.k16_host_unregister.argtypes = [vp, vp]
    L.k16_timer_start.argtypes = [vp]
    L.k16_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.k16_kernel_stats_enable.argtypes = [vp, i32]
    L.k16_kernel_stats_reset.argtypes = [vp]
    L.k16_kernel_stats_get.argtypes = [vp, C.c_char_p, C.POINTER(u64), C.POINTER(C.c_double)]
    L.k16_ctx_set_option.argtypes = [vp, i32, i32]
    L.k16_msm.argtypes = [vp, i32, vp, vp, u64, vp, vp]
    L.k16_msm_host.argtypes = [vp, i32, vp, vp, u64, vp, vp]
    L.k16_msm_enqueue.argtypes = [vp, i32, vp, vp, u64]
    L.k16_msm_finish.argtypes = [vp, vp, vp]
    L.k16_msm_finish_group.argtypes = [vp, i32, vp, vp]
    L.k16_msm_pending.argtypes = [vp]
    L.k16_msm_abort_all.argtypes = [vp]
    L.k16_msm_bases_prepare.argtypes = [vp, i32, vp, u64, vp]
    L.k16_msm_enqueue_prepared.argtypes = [vp, i32, vp, vp, u64]
    L.k16_msm_fixed_base_info.argtypes = [u64, C.POINTER(u32), C.POINTER(u64)]
    L.k16_msm_fixed_base_prepare.argtypes = [vp, i32, vp, u64, vp]
    L.k16_msm_enqueue_fixed_base.argtypes = [vp, i32, vp, vp, u64]
    L.k16_msm_set_window_bits.argtypes = [vp, u32]
    L.k16_msm_set_lane.argtypes = [vp, i32]
    L.k16_points_sum.argtypes = [i32, vp, u64, vp, vp]
    L.k16_msm_zero_row_mask.argtypes = [vp, i32, vp, u64, vp]
    L.k16_msm_set_zero_row_mask.argtypes = [vp, vp]
    L.k16_msm_sort_from_lane.argtypes = [vp, i32, i32]
    L.k16_scalar_classes_create.argtypes = [vp, u64, i32, C.POINTER(vp)]
    L.k16_scalar_classes_destroy.argtypes = [vp]
    L.k16_scalar_classes_destroy.restype = None
    L.k16_scalar_classes_build.argtypes = [vp, vp, vp, u64, C.POINTER(vp), i32, C.c_int64]
    L.k16_scalar_classes_counts.argtypes = [vp, vp, vp]
    L.k16_msm_enqueue_classified.argtypes = [vp, i32, vp, vp, i32]
    L.k16_ntt.argtypes = [vp, vp, u64, u64, i32]
    L.k16_ntt_host.argtypes = [vp, vp, u64, u64, i32]
    L.k16_synth_points.argtypes = [vp, i32, u64, u64, vp]
    L.k16_synth_points_scalars.argtypes = [vp, i32, vp, u64, vp]
    L.k16_field_op_vec.argtypes = [vp, i32, i32, vp, vp, vp, u64]
    L.k16_point_op_vec.argtypes = [vp, i32, i32, vp, vp, vp, u64]
    L.k16_prover_create.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.k16_prover_create_mem.argtypes = [vp, vp, sz, C.POINTER(vp)]
    L.k16_prover_create_shared.argtypes = [vp, vp, C.POINTER(vp)]
    L.k16_prover_destroy.argtypes = [vp]
    L.k16_prover_destroy.restype = None
    L.k16_prover_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
    L.k16_prover_prove_file.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, sz, C.POINTER(C.c_float)]
    L.k16_prover_prove_file_timed.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, sz, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.k16_prover_prove_mem.argtypes = [vp, vp, u64, vp, vp, C.c_char_p, sz, C.POINTER(C.c_float)]
    L.k16_prover_compact_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.k16_prover_prove_compact.argtypes = [vp, u64, vp, vp, C.c_char_p, sz, C.POINTER(C.c_float)]
    L.k16_prover_last_h.argtypes = [vp, vp]
    L.k16_fullprover_prove_mem.argtypes = [vp, vp, u64, C.c_char_p, sz, C.POINTER(i32)]
    L.k16_fullprover_compact_lease.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u32)]
    L.k16_fullprover_prove_compact.argtypes = [vp, vp, u64, C.c_char_p, sz, C.POINTER(i32)]
    L.k16_fullprover_compact_cancel.argtypes = [vp, vp]
    L.k16_prover_warmup_status.argtypes = [vp]
    L.k16_vk_create.argtypes = [vp, vp, vp, vp, vp, vp, u32, C.POINTER(vp)]
    L.k16_vk_destroy.argtypes = [vp]
    L.k16_vk_destroy.restype = None
    L.k16_verify_batch.argtypes = [vp, vp, vp, vp, u64, vp]
    L.k16_verify_coop_gt.argtypes = [vp, vp, vp, vp, u64, vp]
    L.k16_pairing_vec.argtypes = [vp, vp, vp, u64, vp]
    L.k16_msm_sharded_create.argtypes = [C.POINTER(i32), i32, i32, u64, C.POINTER(vp)]
    L.k16_msm_sharded_destroy.argtypes = [vp]
    L.k16_msm_sharded_destroy.restype = None
    L.k16_msm_sharded_count.argtypes = [vp]
    L.k16_msm_sharded_range.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u64)]
    L.k16_msm_sharded_ctx.argtypes = [vp, i32]
    L.k16_msm_sharded_ctx.restype = vp
    L.k16_msm_sharded_last_error.argtypes = [vp]
    L.k16_msm_sharded_last_error.restype = C.c_char_p
    L.k16_msm_sharded_set_bases.argtypes = [vp, vp]
    L.k16_msm_sharded_set_bases_device.argtypes = [vp, i32, vp]
    L.k16_msm_sharded_run.argtypes = [vp, vp, vp, vp]
    L.k16_msm_sharded_run_device.argtypes = [vp, C.POINTER(vp), vp, vp]
    L.k16_msm_sharded_last_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.k16_msm_sharded_set_piece_rows.argtypes = [vp, u64, u64]
    L.k16_rank_comm_unique_id.argtypes = [vp]
    L.k16_rank_comm_load_error.argtypes = []
    L.k16_rank_comm_load_error.restype = C.c_char_p
    L.k16_rank_comm_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.k16_rank_comm_destroy.argtypes = [vp]
    L.k16_rank_comm_destroy.restype = None
    L.k16_rank_comm_allgather_fold.argtypes = [vp, i32, vp, vp, vp]
    L.k16_rank_comm_allgather_start.argtypes = [vp, i32, vp]
    L.k16_rank_comm_allgather_finish.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return a


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._chk(ctx.L.k16_dev_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.L.k16_h2d(self.ctx.h, self.ptr, _p(arr), arr.nbytes))
        return self

    def download(self, dtype=np.uint8, shape=None):
        out = np.empty(self.nbytes, dtype=np.uint8)
        self.ctx._chk(self.ctx.L.k16_d2h(self.ctx.h, _p(out), self.ptr, self.nbytes))
        out = out.view(dtype)
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            self.ctx.L.k16_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    def __init__(self, device=0, stream_offset=0):
        """stream_offset: k16_ctx_create_ex -- placeholder streams created before the context's own (include/k16.h)"""
        self.L = load()
        h = C.c_void_p()
        rc = self.L.k16_ctx_create_ex(device, stream_offset, C.byref(h)) if stream_offset else self.L.k16_ctx_create(device, C.byref(h))
        if rc:
            raise K16Error(rc, "k16_ctx_create(device=%d): no usable HIP device" % device)
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            if not getattr(self, "borrowed", False):   # a ShardedMsm's contexts belong to it
                self.L.k16_ctx_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc:
            raise K16Error(rc, (self.L.k16_last_error(self.h) or b"").decode())

    def sync(self):
        self._chk(self.L.k16_sync(self.h))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 16)).upload(arr)

    def host_register(self, arr):
        """page-lock a numpy array the caller keeps alive (k16_host_register): uploads from it are DMA copies"""
        assert arr.flags["C_CONTIGUOUS"]
        self._chk(self.L.k16_host_register(self.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def host_unregister(self, arr):
        self._chk(self.L.k16_host_unregister(self.h, arr.ctypes.data_as(C.c_void_p)))

    def timer_start(self):
        self._chk(self.L.k16_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.L.k16_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def stats_enable(self, on=True):
        self._chk(self.L.k16_kernel_stats_enable(self.h, int(on)))

    def stats_reset(self):
        self._chk(self.L.k16_kernel_stats_reset(self.h))

    def stats_get(self, name):
        n, ms = C.c_uint64(), C.c_double()
        self._chk(self.L.k16_kernel_stats_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def set_option(self, option, value):
        self._chk(self.L.k16_ctx_set_option(self.h, option, value))

    def set_lane(self, lane):
        self._chk(self.L.k16_msm_set_lane(self.h, lane))

    def set_window_bits(self, c):
        self._chk(self.L.k16_msm_set_window_bits(self.h, c))

    # ---- MSM
    def msm_device(self, group, d_bases, d_scalars, n):
        x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
        self._chk(self.L.k16_msm(self.h, group, d_bases.ptr if d_bases else None,
                                 d_scalars.ptr if d_scalars else None, n, _p(x), _p(a)))
        return x.tobytes(), a.tobytes()

    def msm_enqueue(self, group, d_bases, d_scalars, n):
        self._chk(self.L.k16_msm_enqueue(self.h, group, d_bases.ptr, d_scalars.ptr, n))

    def bases_prepare(self, group, d_bases, n):
        out = self.alloc(max(n * AFF_BYTES[group], 16))
        self._chk(self.L.k16_msm_bases_prepare(self.h, group, d_bases.ptr, n, out.ptr))
        self.sync()
        return out

    def msm_enqueue_prepared(self, group, d_prepared, d_scalars, n):
        self._chk(self.L.k16_msm_enqueue_prepared(self.h, group, d_prepared.ptr, d_scalars.ptr, n))

    def fixed_base_prepare(self, group, d_bases, n):
        """Window tables for a static table (k16_msm_fixed_base_prepare). Returns (DeviceBuffer, c) or (None, 0) when
        this n has no fixed-base mode."""
        c, rows = C.c_uint32(0), C.c_uint64(0)
        self._chk(self.L.k16_msm_fixed_base_info(n, C.byref(c), C.byref(rows)))
        if c.value == 0:
            return None, 0
        out = self.alloc(rows.value * AFF_BYTES[group])
        self._chk(self.L.k16_msm_fixed_base_prepare(self.h, group, d_bases.ptr, n, out.ptr))
        self.sync()
        return out, c.value

    def msm_enqueue_fixed_base(self, group, d_table, d_scalars, n):
        self._chk(self.L.k16_msm_enqueue_fixed_base(self.h, group, d_table.ptr, d_scalars.ptr, n))

    # ---- scalar-class MSM (witness-like scalars)
    def zero_row_mask(self, group, d_rows, n):
        d = self.alloc(max((n + 63) // 64 * 8, 16))
        self._chk(self.L.k16_msm_zero_row_mask(self.h, group, d_rows.ptr, n, d.ptr))
        self.sync()
        return d

    def msm_set_zero_row_mask(self, d_mask):
        self._chk(self.L.k16_msm_set_zero_row_mask(self.h, d_mask.ptr if d_mask is not None else None))

    def msm_sort_from_lane(self, lane, derive=False):
        self._chk(self.L.k16_msm_sort_from_lane(self.h, lane, 1 if derive else 0))

    def classes_create(self, max_n, max_sets=1):
        h = C.c_void_p()
        self._chk(self.L.k16_scalar_classes_create(self.h, max_n, max_sets, C.byref(h)))
        return h

    def classes_destroy(self, cls):
        self.L.k16_scalar_classes_destroy(cls)

    def classes_build(self, cls, d_scalars, n, masks=None, n_wide_bound=-1):
        """masks: list of DeviceBuffer / None, one per set (None: one set, no zero rows)."""
        masks = masks if masks is not None else [None]
        arr = (C.c_void_p * len(masks))(*[(m.ptr if m is not None else None) for m in masks])
        self._chk(self.L.k16_scalar_classes_build(self.h, cls, d_scalars.ptr if d_scalars else None, n, arr, len(masks), n_wide_bound))

    def classes_counts(self, cls, n_sets):
        out = np.zeros(n_sets * 8 + 1, dtype=np.uint32)
        self._chk(self.L.k16_scalar_classes_counts(self.h, cls, _p(out)))
        return out

    def msm_enqueue_classified(self, group, d_prepared, cls, set_index=0):
        self._chk(self.L.k16_msm_enqueue_classified(self.h, group, d_prepared.ptr if d_prepared else None, cls, set_index))

    def msm_finish(self, group):
        x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
        self._chk(self.L.k16_msm_finish_group(self.h, group, _p(x), _p(a)))   # refuses a result of the other group
        return x.tobytes(), a.tobytes()

    def msm_pending(self):
        return self.L.k16_msm_pending(self.h)

    def msm_abort_all(self):
        self._chk(self.L.k16_msm_abort_all(self.h))

    def msm(self, group, bases, scalars):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        n = scalars.shape[0] if scalars.ndim == 2 else 0
        x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
        self._chk(self.L.k16_msm_host(self.h, group, _p(bases), _p(scalars), n, _p(x), _p(a)))
        return x.tobytes(), a.tobytes()

    def synth_points(self, group, start, n):
        """Device buffer with (start+i+1)*G, i < n (affine Montgomery)."""
        d = self.alloc(max(n * AFF_BYTES[group], 16))
        self._chk(self.L.k16_synth_points(self.h, group, start, n, d.ptr))
        self.sync()
        return d

    def synth_points_scalars(self, group, scalars):
        """scalars: list of ints (any size, reduced mod r here) -> uint8 array (n, AFF_BYTES) of scalar_i * G."""
        R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
        n = len(scalars)
        if n == 0:
            return np.zeros((0, AFF_BYTES[group]), dtype=np.uint8)
        buf = np.frombuffer(b"".join((int(k) % R_MOD).to_bytes(32, "little") for k in scalars), dtype=np.uint8)
        d_s = self.to_device(buf)
        d_o = self.alloc(n * AFF_BYTES[group])
        self._chk(self.L.k16_synth_points_scalars(self.h, group, d_s.ptr, n, d_o.ptr))
        self.sync()
        out = d_o.download(np.uint8, (n, AFF_BYTES[group])).copy()
        d_s.free()
        d_o.free()
        return out

    # ---- NTT
    def ntt(self, a, max_domain=None, inverse=False):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        n = a.shape[0]
        self._chk(self.L.k16_ntt_host(self.h, _p(a), n, max_domain or n, 1 if inverse else 0))
        return a

    def ntt_device(self, d_a, n, max_domain=None, inverse=False):
        self._chk(self.L.k16_ntt(self.h, d_a.ptr, n, max_domain or n, 1 if inverse else 0))

    # ---- batch primitives
    def field_op_vec(self, field, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        n = a.shape[0]
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.zeros_like(a)              # (n, 4) u64; (n, 8) for FQ2N elements
        self._chk(self.L.k16_field_op_vec(self.h, field, op, _p(a), _p(b), _p(r), n))
        return r

    def point_op_vec(self, group, op, p1, p2=None):
        p1 = np.ascontiguousarray(p1, dtype=np.uint8)
        n = p1.shape[0]
        if p2 is not None:
            p2 = np.ascontiguousarray(p2, dtype=np.uint8)
        r = np.zeros((n, XYZZ_BYTES[group]), dtype=np.uint8)
        self._chk(self.L.k16_point_op_vec(self.h, group, op, _p(p1), _p(p2), _p(r), n))
        return r


def points_sum(group, parts):
    """parts: uint8 (k, XYZZ_BYTES). Host-side fold of per-shard partial MSM results."""
    L = load()
    parts = np.ascontiguousarray(parts, dtype=np.uint8)
    x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
    a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
    rc = L.k16_points_sum(group, _p(parts), parts.shape[0], _p(x), _p(a))
    if rc:
        raise K16Error(rc)
    return x.tobytes(), a.tobytes()


class Prover:
    """Mirror of the reference's FullProver(zkey).prove(wtns) (fullprover.hpp:52-64) over the C ABI."""

    def __init__(self, ctx, zkey_path, share_key_of=None):
        """share_key_of: another Prover of the same key on the same device -- k16_prover_create_shared (one resident key)"""
        self.ctx = ctx
        h = C.c_void_p()
        if share_key_of is not None:
            rc = ctx.L.k16_prover_create_shared(ctx.h, share_key_of.h, C.byref(h))
        else:
            rc = ctx.L.k16_prover_create(ctx.h, zkey_path.encode(), C.byref(h))
        if rc:
            raise K16Error(rc, (ctx.L.k16_last_error(ctx.h) or b"").decode())
        self.h = h

    def info(self):
        nv, npub, ds, nc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        self.ctx._chk(self.ctx.L.k16_prover_info(self.h, C.byref(nv), C.byref(npub), C.byref(ds), C.byref(nc)))
        return dict(n_vars=nv.value, n_public=npub.value, domain_size=ds.value, n_coefs=nc.value)

    def prove_file(self, wtns_path, r=None, s=None):
        buf = C.create_string_buffer(4096)
        ms = C.c_float()
        R_ = np.frombuffer(bytes(r), dtype=np.uint8).copy() if r is not None else None
        S_ = np.frombuffer(bytes(s), dtype=np.uint8).copy() if s is not None else None
        rc = self.ctx.L.k16_prover_prove_file(self.h, wtns_path.encode(), _p(R_), _p(S_), buf, 4096, C.byref(ms))
        if rc < 0:
            raise K16Error(rc, (self.ctx.L.k16_last_error(self.ctx.h) or b"").decode())
        self.last_device_ms = ms.value
        return buf.value.decode()

    def prove_mem(self, wtns, r=None, s=None):
        wtns = np.ascontiguousarray(wtns, dtype=np.uint8)
        n_vars = wtns.size // 32
        buf = C.create_string_buffer(4096)
        ms = C.c_float()
        R_ = np.frombuffer(bytes(r), dtype=np.uint8).copy() if r is not None else None
        S_ = np.frombuffer(bytes(s), dtype=np.uint8).copy() if s is not None else None
        rc = self.ctx.L.k16_prover_prove_mem(self.h, _p(wtns), n_vars, _p(R_), _p(S_), buf, 4096, C.byref(ms))
        if rc < 0:
            raise K16Error(rc, (self.ctx.L.k16_last_error(self.ctx.h) or b"").decode())
        self.last_device_ms = ms.value
        return buf.value.decode()

    def compact_buffers(self):
        """The prover's pinned upload buffers as numpy views (k16_prover_compact_buffers): narrow[n_vars] uint8,
        wide_idx[cap] uint32, wide_val[cap, 32] uint8."""
        a, b, c, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        rc = self.ctx.L.k16_prover_compact_buffers(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(cap))
        if rc < 0:
            raise K16Error(rc, (self.ctx.L.k16_last_error(self.ctx.h) or b"").decode())
        n, k = self.info()["n_vars"], int(cap.value)
        narrow = np.ctypeslib.as_array((C.c_uint8 * n).from_address(a.value))
        idx = np.ctypeslib.as_array((C.c_uint32 * k).from_address(b.value))
        val = np.ctypeslib.as_array((C.c_uint8 * (k * 32)).from_address(c.value)).reshape(k, 32)
        return narrow, idx, val

    def prove_compact(self, n_wide, r=None, s=None):
        """k16_prover_prove_compact: the witness is what compact_buffers() holds."""
        buf = C.create_string_buffer(4096)
        ms = C.c_float()
        R_ = np.frombuffer(bytes(r), dtype=np.uint8).copy() if r is not None else None
        S_ = np.frombuffer(bytes(s), dtype=np.uint8).copy() if s is not None else None
        rc = self.ctx.L.k16_prover_prove_compact(self.h, int(n_wide), _p(R_), _p(S_), buf, 4096, C.byref(ms))
        if rc < 0:
            raise K16Error(rc, (self.ctx.L.k16_last_error(self.ctx.h) or b"").decode())
        self.last_device_ms = ms.value
        return buf.value.decode()

    def warmup_status(self):
        return int(self.ctx.L.k16_prover_warmup_status(self.h))

    def last_h(self):
        n = self.info()["domain_size"]
        out = np.zeros((n, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.k16_prover_last_h(self.h, _p(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.L.k16_prover_destroy(self.h)
            self.h = None


class VerifyingKey:
    """Groth16 verifying key resident on the GPU (k16_vk_create) + batched verification (k16_verify_batch): the mirror of
    the service's prepared_vk / verify_proof pair (prover-service/src/request_handler/types.rs:141-196,
    prover_handler.rs:329-336).  vk: dict(alpha1, beta2, gamma2, delta2, ic=[...]) of affine Montgomery bytes."""

    def __init__(self, ctx, vk):
        self.ctx, self.n_ic = ctx, len(vk["ic"])
        h = C.c_void_p()
        b = lambda x: np.frombuffer(bytes(x), dtype=np.uint8).copy()
        ic = b(b"".join(vk["ic"]))
        ctx._chk(ctx.L.k16_vk_create(ctx.h, _p(b(vk["alpha1"])), _p(b(vk["beta2"])), _p(b(vk["gamma2"])), _p(b(vk["delta2"])),
                                     _p(ic), self.n_ic, C.byref(h)))
        self.h = h

    def verify_batch(self, proofs, inputs):
        """proofs: list of 256-byte A | B | C; inputs: per proof a list of n_ic - 1 ints.  Returns a list of bools."""
        n = len(proofs)
        if n == 0:
            return []
        pr = np.frombuffer(b"".join(bytes(p) for p in proofs), dtype=np.uint8).copy()
        assert pr.size == 256 * n
        inp = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for row in inputs for x in row), dtype=np.uint8).copy()
        assert inp.size == n * (self.n_ic - 1) * 32
        ok = np.zeros(n, dtype=np.uint8)
        self.ctx._chk(self.ctx.L.k16_verify_batch(self.ctx.h, self.h, _p(pr), _p(inp) if inp.size else None, n, _p(ok)))
        return [bool(v) for v in ok]

    def coop_gt(self, proofs, inputs):
        """The GT value e(A,B) e(vk_x,-gamma) e(C,-delta) of every proof as the wave-cooperative path computes it
        (k16_verify_coop_gt): (n, 384) uint8.  Raises K16Error(ARG) when that path does not apply."""
        n = len(proofs)
        pr = np.frombuffer(b"".join(bytes(p) for p in proofs), dtype=np.uint8).copy()
        inp = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for row in inputs for x in row), dtype=np.uint8).copy()
        out = np.zeros((n, 384), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.k16_verify_coop_gt(self.ctx.h, self.h, _p(pr), _p(inp) if inp.size else None, n, _p(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.L.k16_vk_destroy(self.h)
            self.h = None


def pairing_vec(ctx, g1, g2):
    """e(P_i, Q_i) for affine Montgomery G1 (n, 64) / G2 (n, 128) arrays -> (n, 384) uint8 (k16_pairing_vec)."""
    g1 = np.ascontiguousarray(g1, dtype=np.uint8)
    g2 = np.ascontiguousarray(g2, dtype=np.uint8)
    n = g1.shape[0]
    out = np.zeros((n, 384), dtype=np.uint8)
    ctx._chk(ctx.L.k16_pairing_vec(ctx.h, _p(g1), _p(g2), n, _p(out)))
    return out


class ShardedMsm:
    """One MSM sharded over several devices from ONE process (include/k16.h: k16_msm_sharded_*): a context per entry of
    `devices` (entries may repeat), contiguous shards, host-side EC-add fold of the per-shard partial results."""

    def __init__(self, devices, group, n):
        self.L = load()
        self.group, self.n = group, n
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.L.k16_msm_sharded_create(arr, len(devices), group, n, C.byref(h))
        if rc:
            raise K16Error(rc, "k16_msm_sharded_create")
        self.h = h

    def _chk(self, rc):
        if rc:
            raise K16Error(rc, (self.L.k16_msm_sharded_last_error(self.h) or b"").decode())

    def count(self):
        return self.L.k16_msm_sharded_count(self.h)

    def shard_range(self, r):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._chk(self.L.k16_msm_sharded_range(self.h, r, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def shard_ctx(self, r):
        """A borrowed Context over shard r's k16_ctx (owned by the ShardedMsm: do not close it)."""
        c = Context.__new__(Context)
        c.L, c.h, c.borrowed = self.L, C.c_void_p(self.L.k16_msm_sharded_ctx(self.h, r)), True
        c.device = -1
        return c

    def set_bases(self, h_bases):
        h_bases = np.ascontiguousarray(h_bases)
        assert h_bases.nbytes == self.n * AFF_BYTES[self.group]
        self._chk(self.L.k16_msm_sharded_set_bases(self.h, _p(h_bases)))

    def set_bases_device(self, r, d_slice):
        self._chk(self.L.k16_msm_sharded_set_bases_device(self.h, r, d_slice.ptr if d_slice is not None else None))

    def run(self, h_scalars):
        h_scalars = np.ascontiguousarray(h_scalars)
        assert h_scalars.nbytes == self.n * 32
        x = np.zeros(XYZZ_BYTES[self.group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[self.group], dtype=np.uint8)
        self._chk(self.L.k16_msm_sharded_run(self.h, _p(h_scalars), _p(x), _p(a)))
        return x.tobytes(), a.tobytes()

    def run_device(self, d_scalars):
        arr = (C.c_void_p * len(d_scalars))(*[d.ptr for d in d_scalars])
        x = np.zeros(XYZZ_BYTES[self.group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[self.group], dtype=np.uint8)
        self._chk(self.L.k16_msm_sharded_run_device(self.h, arr, _p(x), _p(a)))
        return x.tobytes(), a.tobytes()

    def set_piece_rows(self, host_rows=0, device_rows=0):
        """rows per device pass inside a shard (include/k16.h; 0 = default 2^22 / 2^24); same result for any value"""
        self._chk(self.L.k16_msm_sharded_set_piece_rows(self.h, host_rows, device_rows))

    def last_ms(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.k16_msm_sharded_last_ms(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(shards_ms=a.value, fold_ms=b.value, total_ms=c.value)

    def close(self):
        if self.h:
            self.L.k16_msm_sharded_destroy(self.h)
            self.h = None


class RankComm:
    """One process per GPU: the shards' partial results exchanged with ONE ncclAllGather (RCCL, dlopen'ed by the library) and
    folded on every rank (include/k16.h: k16_rank_comm_*).  `unique_id` = the 128 bytes rank 0 got from RankComm.unique_id(),
    handed to the other ranks by the launcher."""

    @staticmethod
    def unique_id():
        L = load()
        buf = np.zeros(128, dtype=np.uint8)
        rc = L.k16_rank_comm_unique_id(_p(buf))
        if rc:
            raise K16Error(rc, (L.k16_rank_comm_load_error() or b"").decode())
        return buf.tobytes()

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        uid = np.frombuffer(unique_id, dtype=np.uint8).copy()
        ctx._chk(self.L.k16_rank_comm_create(ctx.h, rank, world, _p(uid), C.byref(h)))
        self.h, self.world = h, world

    def allgather_fold(self, group, partial_xyzz):
        part = np.frombuffer(partial_xyzz, dtype=np.uint8).copy()
        x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
        self.ctx._chk(self.L.k16_rank_comm_allgather_fold(self.h, group, _p(part), _p(x), _p(a)))
        return x.tobytes(), a.tobytes()

    def allgather_start(self, group, partial_xyzz):
        """enqueue one exchange and return (include/k16.h: up to 4 in flight, completed in order by allgather_finish)"""
        part = np.frombuffer(partial_xyzz, dtype=np.uint8).copy()
        self.ctx._chk(self.L.k16_rank_comm_allgather_start(self.h, group, _p(part)))
        self._groups = getattr(self, "_groups", []) + [group]

    def allgather_finish(self):
        group = self._groups.pop(0)
        x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
        a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
        self.ctx._chk(self.L.k16_rank_comm_allgather_finish(self.h, _p(x), _p(a)))
        return x.tobytes(), a.tobytes()

    def close(self):
        if self.h:
            self.L.k16_rank_comm_destroy(self.h)
            self.h = None
