"""Builds synthetic Groth16 .zkey / .wtns files in the iden3 binary container format the prover parses
(rust-rapidsnark/rapidsnark/src/binfile_utils.cpp:13-58, zkey_utils.hpp:49-87,
wtns_utils.hpp:29-44).  The "circuit" is random: the proof will not verify, but prove() is a deterministic
function of (zkey, witness, r, s), which is what the parity tests compare between the HIP path and the oracle.
"""
import struct

import numpy as np

import oracle_lib as ol
import pymodel as pm


def _section(t, payload):
    return struct.pack("<IQ", t, len(payload)) + payload


def build_zkey(path, n_vars, n_public, domain_size, n_coefs, seed=1, zero_frac=0.3, long_rows=()):
    """long_rows: lengths of extra constraint rows of one matrix each (65 ... thousands of entries, as circom's Num2Bits or
    a wide linear combination produce); their coefficients are part of n_coefs and keep the file sorted by constraint."""
    rs = np.random.RandomState(seed)
    g1 = ol.gen_points(ol.G1, 100, 6)          # alpha1, beta1, delta1 + spare
    g2 = ol.gen_points(ol.G2, 50, 3)           # beta2, gamma2, delta2
    hdr = struct.pack("<I", 32) + pm.limbs(pm.Q) + struct.pack("<I", 32) + pm.limbs(pm.R)
    hdr += struct.pack("<III", n_vars, n_public, domain_size)
    hdr += bytes(g1[0]) + bytes(g1[1]) + bytes(g2[0]) + bytes(g2[1]) + bytes(g1[2]) + bytes(g2[2])
    # coefficients: (m, c, s, value * R^2 mod r), grouped by c then m as snarkjs writes them
    m = rs.randint(0, 2, size=n_coefs).astype(np.uint32)
    c = rs.randint(0, domain_size, size=n_coefs).astype(np.uint32)
    at = 0
    for k, ln in enumerate(long_rows):          # one (m, c) row gets ln entries
        assert at + ln <= n_coefs
        c[at:at + ln] = (k * 7919 + 5) % domain_size
        m[at:at + ln] = k & 1
        at += ln
    order = np.argsort(c, kind="stable")
    c, m = c[order], m[order]
    s = rs.randint(0, n_vars, size=n_coefs).astype(np.uint32)
    r2 = pow(pm.MONT, 2, pm.R)
    coefs = bytearray(struct.pack("<I", n_coefs))
    for i in range(n_coefs):
        v = int(rs.randint(1, 1 << 30)) if rs.rand() < 0.7 else pm.SplitMix64(seed * 7919 + i).below(pm.R)
        coefs += struct.pack("<III", int(m[i]), int(c[i]), int(s[i])) + pm.limbs(v * r2 % pm.R)

    def pts(group, start, n, zf):
        p = ol.gen_points(group, start, n)
        if n and zf > 0:
            p[rs.rand(n) < zf] = 0              # sparse B1/B2 columns are (0,0) in real keys
        return p.tobytes()

    secs = [
        _section(1, struct.pack("<I", 1)),
        _section(2, hdr),
        _section(3, bytes(ol.gen_points(ol.G1, 7, n_public + 1).tobytes())),
        _section(4, bytes(coefs)),
        _section(5, pts(ol.G1, 1000, n_vars, 0.0)),
        _section(6, pts(ol.G1, 200000, n_vars, zero_frac)),
        _section(7, pts(ol.G2, 3000, n_vars, zero_frac)),
        _section(8, pts(ol.G1, 400000, n_vars - n_public - 1, 0.0)),
        _section(9, pts(ol.G1, 600000, domain_size, 0.0)),
    ]
    with open(path, "wb") as f:
        f.write(b"zkey" + struct.pack("<II", 1, len(secs)) + b"".join(secs))


def build_wtns(path, n_vars, seed=2):
    """90 % bits, 8 % bytes, 2 % full-width field elements; w[0] = 1."""
    rs = np.random.RandomState(seed)
    w = np.zeros((n_vars, 32), dtype=np.uint8)
    u = rs.rand(n_vars)
    w[u < 0.90, 0] = rs.randint(0, 2, size=int((u < 0.90).sum()))
    byts = (u >= 0.90) & (u < 0.98)
    w[byts, 0] = rs.randint(0, 256, size=int(byts.sum()))
    full = np.nonzero(u >= 0.98)[0]
    for k, i in enumerate(full):
        w[i] = np.frombuffer(pm.limbs(pm.SplitMix64(seed * 104729 + k).below(pm.R)), dtype=np.uint8)
    w[0] = 0
    w[0, 0] = 1
    sec1 = struct.pack("<I", 32) + pm.limbs(pm.R) + struct.pack("<I", n_vars)
    with open(path, "wb") as f:
        f.write(b"wtns" + struct.pack("<II", 2, 2) + _section(1, sec1) + _section(2, w.tobytes()))
    return w
