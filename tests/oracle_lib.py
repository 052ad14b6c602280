"""ctypes binding of the CPU oracle (oracle/liboracle_bn254.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_bn254.so")

OP_ADD, OP_SUB, OP_NEG, OP_MUL, OP_SQR, OP_TOMONT, OP_FROMMONT, OP_INV = range(8)
PT_ADD, PT_MADD, PT_DBL, PT_NEG = range(4)
FQ, FR = 0, 1
G1, G2 = 0, 1

AFF_BYTES = {G1: 64, G2: 128}
XYZZ_BYTES = {G1: 128, G2: 256}

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
        L.ora_field_op.argtypes = [i32, i32, vp, vp, vp]
        L.ora_field_op_vec.argtypes = [i32, i32, vp, vp, vp, u64]
        L.ora_fe_to_dec.argtypes = [i32, vp, C.c_char_p]
        L.ora_fe_to_dec.restype = i32
        L.ora_fe_from_dec.argtypes = [i32, C.c_char_p, vp]
        L.ora_fq2_op.argtypes = [i32, vp, vp, vp]
        L.ora_pt_op.argtypes = [i32, i32, vp, vp, vp]
        L.ora_pt_to_affine.argtypes = [i32, vp, vp]
        L.ora_pt_eq.argtypes = [i32, vp, vp]
        L.ora_pt_eq.restype = i32
        L.ora_generator.argtypes = [i32, vp]
        L.ora_mul_scalar.argtypes = [i32, vp, vp, u32, vp]
        L.ora_gen_points.argtypes = [i32, u64, u64, vp]
        L.ora_msm.argtypes = [i32, vp, vp, u64, u64, i32, vp, vp]
        L.ora_ntt.argtypes = [vp, u64, u64, i32]
        L.ora_ntt.restype = i32
        L.ora_ntt_root.argtypes = [u64, u32, u64, vp]
        L.ora_ntt_root.restype = i32
        L.ora_zkey_info.argtypes = [C.c_char_p, vp, vp, vp, vp]
        L.ora_zkey_info.restype = i32
        L.ora_prove_files.argtypes = [C.c_char_p, C.c_char_p, vp, vp, i32, C.c_char_p, C.c_size_t, vp]
        L.ora_prove_files.restype = i32
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def field_op(field, op, a, b=None):
    """a, b: bytes (32 B each). Returns 32 bytes."""
    A = np.frombuffer(bytes(a), dtype=np.uint64).copy()
    B = np.frombuffer(bytes(b), dtype=np.uint64).copy() if b is not None else None
    Rr = np.zeros(4, dtype=np.uint64)
    lib().ora_field_op(field, op, _ptr(A), _ptr(B), _ptr(Rr))
    return Rr.tobytes()


def field_op_vec(field, op, a, b=None):
    """a, b: uint64 arrays of shape (n,4)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n = a.shape[0]
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint64)
    r = np.zeros((n, 4), dtype=np.uint64)
    lib().ora_field_op_vec(field, op, _ptr(a), _ptr(b), _ptr(r), n)
    return r


def fe_to_dec(field, a):
    A = np.frombuffer(bytes(a), dtype=np.uint64).copy()
    buf = C.create_string_buffer(100)
    lib().ora_fe_to_dec(field, _ptr(A), buf)
    return buf.value.decode()


def fe_from_dec(field, s):
    Rr = np.zeros(4, dtype=np.uint64)
    lib().ora_fe_from_dec(field, s.encode(), _ptr(Rr))
    return Rr.tobytes()


def fq2_op(op, a, b=None):
    A = np.frombuffer(bytes(a), dtype=np.uint64).copy()
    B = np.frombuffer(bytes(b), dtype=np.uint64).copy() if b is not None else None
    Rr = np.zeros(8, dtype=np.uint64)
    lib().ora_fq2_op(op, _ptr(A), _ptr(B), _ptr(Rr))
    return Rr.tobytes()


def pt_op(group, op, p1, p2=None):
    A = np.frombuffer(bytes(p1), dtype=np.uint8).copy()
    B = np.frombuffer(bytes(p2), dtype=np.uint8).copy() if p2 is not None else None
    Rr = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
    lib().ora_pt_op(group, op, _ptr(A), _ptr(B), _ptr(Rr))
    return Rr.tobytes()


def pt_to_affine(group, p):
    A = np.frombuffer(bytes(p), dtype=np.uint8).copy()
    Rr = np.zeros(AFF_BYTES[group], dtype=np.uint8)
    lib().ora_pt_to_affine(group, _ptr(A), _ptr(Rr))
    return Rr.tobytes()


def pt_eq(group, p1, p2):
    A = np.frombuffer(bytes(p1), dtype=np.uint8).copy()
    B = np.frombuffer(bytes(p2), dtype=np.uint8).copy()
    return bool(lib().ora_pt_eq(group, _ptr(A), _ptr(B)))


def generator(group):
    Rr = np.zeros(AFF_BYTES[group], dtype=np.uint8)
    lib().ora_generator(group, _ptr(Rr))
    return Rr.tobytes()


def mul_scalar(group, base_aff, scalar):
    A = np.frombuffer(bytes(base_aff), dtype=np.uint8).copy()
    S = np.frombuffer(bytes(scalar), dtype=np.uint8).copy()
    Rr = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
    lib().ora_mul_scalar(group, _ptr(A), _ptr(S), len(S), _ptr(Rr))
    return Rr.tobytes()


def gen_points(group, start, n):
    """(start+i+1)*G for i in [0,n) as a uint8 array (n, AFF_BYTES)."""
    out = np.zeros((n, AFF_BYTES[group]), dtype=np.uint8)
    lib().ora_gen_points(group, start, n, _ptr(out))
    return out


def msm(group, bases, scalars, nthreads=1):
    """bases: uint8 (n, AFF_BYTES); scalars: uint8 (n, 32). Returns (xyzz bytes, affine bytes)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    n = scalars.shape[0] if scalars.ndim == 2 else 0
    x = np.zeros(XYZZ_BYTES[group], dtype=np.uint8)
    a = np.zeros(AFF_BYTES[group], dtype=np.uint8)
    lib().ora_msm(group, _ptr(bases), _ptr(scalars), 32, n, nthreads, _ptr(x), _ptr(a))
    return x.tobytes(), a.tobytes()


def ntt(a, max_domain=None, inverse=False):
    """a: uint64 (n,4) Montgomery. Returns transformed copy."""
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    n = a.shape[0]
    rc = lib().ora_ntt(_ptr(a), n, max_domain or n, 1 if inverse else 0)
    if rc:
        raise RuntimeError("ora_ntt rc=%d" % rc)
    return a


def ntt_root(max_domain, domain_pow, idx):
    Rr = np.zeros(4, dtype=np.uint64)
    rc = lib().ora_ntt_root(max_domain, domain_pow, idx, _ptr(Rr))
    if rc:
        raise RuntimeError("ora_ntt_root rc=%d" % rc)
    return Rr.tobytes()


def zkey_info(path):
    nv, npub, ds = C.c_uint32(), C.c_uint32(), C.c_uint32()
    nc = C.c_uint64()
    rc = lib().ora_zkey_info(path.encode(), C.byref(nv), C.byref(npub), C.byref(ds), C.byref(nc))
    if rc:
        raise RuntimeError("ora_zkey_info rc=%d" % rc)
    return dict(n_vars=nv.value, n_public=npub.value, domain_size=ds.value, n_coefs=nc.value)


def prove_files(zkey, wtns, r=b"\0" * 32, s=b"\0" * 32, nthreads=1, want_h=False):
    R_ = np.frombuffer(bytes(r), dtype=np.uint8).copy()
    S_ = np.frombuffer(bytes(s), dtype=np.uint8).copy()
    buf = C.create_string_buffer(4096)
    h = None
    if want_h:
        h = np.zeros((zkey_info(zkey)["domain_size"], 4), dtype=np.uint64)
    rc = lib().ora_prove_files(zkey.encode(), wtns.encode(), _ptr(R_), _ptr(S_), nthreads, buf, 4096, _ptr(h))
    if rc < 0:
        raise RuntimeError("ora_prove_files rc=%d" % rc)
    js = buf.value.decode()
    return (js, h) if want_h else js


# ---------------------------------------------------------------- pairing / Groth16 verification (oracle/pairing_ref.h)
GT_BYTES = 384


def _pairing_lib():
    L = lib()
    if not getattr(L, "_pairing_ready", False):
        vp = C.c_void_p
        L.ora_miller.argtypes = [vp, vp, vp]
        L.ora_miller.restype = None
        L.ora_pairing.argtypes = [vp, vp, vp]
        L.ora_pairing.restype = C.c_int
        L.ora_final_exp.argtypes = [vp, vp]
        L.ora_final_exp.restype = C.c_int
        L.ora_gt_mul.argtypes = [vp, vp, vp]
        L.ora_gt_mul.restype = None
        L.ora_groth16_verify.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, vp]
        L.ora_groth16_verify.restype = C.c_int
        L._pairing_ready = True
    return L


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def miller(g1_aff, g2_aff):
    out = np.zeros(GT_BYTES, dtype=np.uint8)
    _pairing_lib().ora_miller(_ptr(_u8(g1_aff)), _ptr(_u8(g2_aff)), _ptr(out))
    return out.tobytes()


def pairing(g1_aff, g2_aff):
    out = np.zeros(GT_BYTES, dtype=np.uint8)
    rc = _pairing_lib().ora_pairing(_ptr(_u8(g1_aff)), _ptr(_u8(g2_aff)), _ptr(out))
    if rc:
        raise RuntimeError("ora_pairing rc=%d" % rc)
    return out.tobytes()


def final_exp(f):
    out = np.zeros(GT_BYTES, dtype=np.uint8)
    rc = _pairing_lib().ora_final_exp(_ptr(_u8(f)), _ptr(out))
    if rc:
        raise RuntimeError("ora_final_exp rc=%d" % rc)
    return out.tobytes()


def gt_mul(a, b):
    out = np.zeros(GT_BYTES, dtype=np.uint8)
    _pairing_lib().ora_gt_mul(_ptr(_u8(a)), _ptr(_u8(b)), _ptr(out))
    return out.tobytes()


def groth16_verify(vk, proof, inputs):
    """vk: dict(alpha1, beta2, gamma2, delta2, ic=[...]) of affine Montgomery bytes (groth16_io.vk_from_json);
    proof: 256 B A | B | C; inputs: list of ints.  Returns True / False."""
    ic = _u8(b"".join(vk["ic"]))
    inp = _u8(b"".join(int(x).to_bytes(32, "little") for x in inputs)) if inputs else None
    rc = _pairing_lib().ora_groth16_verify(_ptr(_u8(vk["alpha1"])), _ptr(_u8(vk["beta2"])), _ptr(_u8(vk["gamma2"])),
                                           _ptr(_u8(vk["delta2"])), _ptr(ic), len(vk["ic"]), _ptr(_u8(proof)), _ptr(inp))
    if rc < 0:
        raise RuntimeError("ora_groth16_verify rc=%d" % rc)
    return rc == 1
