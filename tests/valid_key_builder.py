"""Builds a VALID synthetic Groth16 key of any size -- circuit, trusted set-up from a known trapdoor, verification key,
satisfying witness -- so that proofs of non-toy circuits can be checked the way the reference checks its proofs: they
must VERIFY (prover-service/src/tests/prover_handler.rs:279-290, prover_handler.rs:329-336), not only equal the oracle's
bytes.  The only verifying fixture the reference ships is the 3-wire toy circuit; the real Keyless key is not available
offline.

Circuit (R1CS over Fr), shaped like the Keyless witness (mostly bits, some bytes, a few full-width values):
    wire 0 = 1, wire 1 = the public input, then n_bits bit wires, n_bytes free byte wires, n_prod product wires
    bit wire i      : w_i * (w_i - 1) = 0                A = e_i, B = e_i - e_0, C = 0
    product wire c  : (k1 w_a + k2 w_b) * (k3 w_d) = w_c   with a, b, d earlier wires and small k
    and the nPublic + 1 rows snarkjs appends (A = e_i for the public wires, B = C = 0)
Set-up with trapdoor (tau, alpha, beta, gamma, delta), exactly what the prover consumes:
    A_i = [a_i(tau)]_1, B1_i = [b_i(tau)]_1, B2_i = [b_i(tau)]_2, C_i = [(beta a_i + alpha b_i + c_i)/delta]_1 (private wires),
    IC_i = [(beta a_i + alpha b_i + c_i)/gamma]_1 (public wires),
    H_i = [Z(tau) l_i(tau) / (Z(x_i) delta)]_1 with l_i the Lagrange basis of the odd coset x_i = g^(2i+1), g the primitive
    2N-th root the prover's FFT uses (5^((r-1)/2N)), Z(x_i) = -2: the prover multiplies them with h_i = (A.B - C)(x_i)
    (RS/groth16.cpp:172-283), so sum h_i H_i = [H(tau) Z(tau) / delta]_1.
`points(group, scalars)` supplies scalar * generator for lists of scalars: the GPU (k16.Context.synth_points_scalars) for
large keys, the oracle for small ones.
"""
import struct

import numpy as np

import pymodel as pm

R = pm.R


def _inv(x):
    return pow(x, -1, R)


def _batch_inv(v):
    pref, acc = [0] * len(v), 1
    for i, x in enumerate(v):
        pref[i] = acc
        acc = acc * x % R
    inv = _inv(acc)
    out = [0] * len(v)
    for i in range(len(v) - 1, -1, -1):
        out[i] = inv * pref[i] % R
        inv = inv * v[i] % R
    return out


def _lagrange_at(t, N, omega):
    """[L_j(t) for j < N] over the domain {omega^j}: (t^N - 1)/N * omega^j / (t - omega^j)."""
    pw, u = [1] * N, 1
    for j in range(N):
        pw[j] = u
        u = u * omega % R
    dinv = _batch_inv([(t - x) % R for x in pw])
    c = (pow(t, N, R) - 1) * _inv(N) % R
    return [c * pw[j] % R * dinv[j] % R for j in range(N)]


def _section(t, payload):
    return struct.pack("<IQ", t, len(payload)) + payload


def build(points, n_bits, n_bytes, n_prod, seed=1):
    """Returns dict(zkey=bytes, vk=dict of affine Montgomery bytes (groth16_io format), witness=(n_vars, 32) uint8,
    public=[int], n_vars, n_public, domain, n_coefs)."""
    rng = pm.SplitMix64(seed)                 # circuit structure and trapdoor
    n_public = 1
    n_vars = 2 + n_bits + n_bytes + n_prod
    M = n_bits + n_prod
    N = 4
    while N < M + n_public + 1:
        N *= 2
    bit0, byte0, prod0 = 2, 2 + n_bits, 2 + n_bits + n_bytes
    # ---- constraints: per row lists of (wire, coef)
    rowsA, rowsB, rowsC, prods = [], [], [], []
    for i in range(bit0, byte0):
        rowsA.append([(i, 1)])
        rowsB.append([(i, 1), (0, R - 1)])
        rowsC.append([])
    for c in range(prod0, n_vars):
        a, b, d = 1 + rng.next() % (c - 1), 1 + rng.next() % (c - 1), 1 + rng.next() % (c - 1)
        k1, k2, k3 = 1 + rng.next() % 1000, 1 + rng.next() % 1000, 1 + rng.next() % 1000
        prods.append((c, a, b, d, k1, k2, k3))
        rowsA.append([(a, k1), (b, k2)] if a != b else [(a, (k1 + k2) % R)])
        rowsB.append([(d, k3)])
        rowsC.append([(c, 1)])
    for i in range(n_public + 1):       # snarkjs: one extra row per public wire (and the constant)
        rowsA.append([(i, 1)])
        rowsB.append([])
        rowsC.append([])
    assert len(rowsA) == M + n_public + 1 <= N

    def make_witness(wseed):
        """A satisfying assignment (ints) for this circuit: free wires from wseed, product wires computed."""
        rw = pm.SplitMix64(wseed)
        w = [0] * n_vars
        w[0] = 1
        w[1] = rw.next() | 1
        for i in range(bit0, byte0):
            w[i] = rw.next() & 1
        for i in range(byte0, prod0):
            w[i] = rw.next() & 0xFF
        for c, a, b, d, k1, k2, k3 in prods:
            w[c] = (k1 * w[a] + k2 * w[b]) % R * (k3 * w[d] % R) % R
        return w

    def witness_bytes(w):
        return np.frombuffer(b"".join(pm.limbs(x) for x in w), dtype=np.uint8).reshape(n_vars, 32).copy()

    w = make_witness(seed * 1000003 + 1)
    if n_vars <= 4096:                     # small keys: check every row (large ones: the proof verifying is the check)
        for ra, rb, rc in zip(rowsA, rowsB, rowsC):
            dot = lambda row: sum(k * w[s] for s, k in row) % R
            assert dot(ra) * dot(rb) % R == dot(rc)
    # ---- trapdoor and QAP evaluations at tau
    tau, alpha, beta, gamma, delta = (1 + rng.below(R - 1) for _ in range(5))
    S = N.bit_length()                       # log2(2N)
    g = pow(5, (R - 1) >> S, R)              # primitive 2N-th root (RS/fft.cpp:60-97: nqr = 5)
    omega = g * g % R
    L = _lagrange_at(tau, N, omega)
    a_t, b_t, c_t = [0] * n_vars, [0] * n_vars, [0] * n_vars
    coefs = []
    r2 = pow(pm.MONT, 2, R)
    for j, (ra, rb, rc) in enumerate(zip(rowsA, rowsB, rowsC)):
        for s, k in ra:
            a_t[s] = (a_t[s] + k * L[j]) % R
            coefs.append((0, j, s, k))
        for s, k in rb:
            b_t[s] = (b_t[s] + k * L[j]) % R
            coefs.append((1, j, s, k))
        for s, k in rc:
            c_t[s] = (c_t[s] + k * L[j]) % R
    z_tau = (pow(tau, N, R) - 1) % R
    # H_i = Z(tau) * l_i(tau) / (-2 delta),  l_i = Lagrange basis of the coset g * {omega^i}: L_i(tau / g)
    Lc = _lagrange_at(tau * _inv(g) % R, N, omega)
    hk = z_tau * _inv((R - 2) * delta % R) % R
    s_h = [hk * x % R for x in Lc]
    dinv, ginv = _inv(delta), _inv(gamma)
    mix = [(beta * a_t[i] + alpha * b_t[i] + c_t[i]) % R for i in range(n_vars)]
    s_c = [mix[i] * dinv % R for i in range(n_public + 1, n_vars)]
    s_ic = [mix[i] * ginv % R for i in range(n_public + 1)]
    # ---- points
    pa = points(0, a_t)
    pb1 = points(0, b_t)
    pb2 = points(1, b_t)
    pc = points(0, s_c)
    ph = points(0, s_h)
    pic = points(0, s_ic)
    hdr1 = points(0, [alpha, beta, delta])
    hdr2 = points(1, [beta, gamma, delta])
    hdr = struct.pack("<I", 32) + pm.limbs(pm.Q) + struct.pack("<I", 32) + pm.limbs(R) + struct.pack("<III", n_vars, n_public, N)
    hdr += bytes(hdr1[0]) + bytes(hdr1[1]) + bytes(hdr2[0]) + bytes(hdr2[1]) + bytes(hdr1[2]) + bytes(hdr2[2])
    cf = np.zeros(len(coefs), dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "V32")])
    cf["m"] = [x[0] for x in coefs]
    cf["c"] = [x[1] for x in coefs]
    cf["s"] = [x[2] for x in coefs]
    cf["v"] = np.frombuffer(b"".join(pm.limbs(x[3] * r2 % R) for x in coefs), dtype="V32")
    secs = [_section(1, struct.pack("<I", 1)), _section(2, hdr), _section(3, pic.tobytes()),
            _section(4, struct.pack("<I", len(coefs)) + cf.tobytes()),
            _section(5, pa.tobytes()), _section(6, pb1.tobytes()), _section(7, pb2.tobytes()),
            _section(8, pc.tobytes()), _section(9, ph.tobytes())]
    zkey = b"zkey" + struct.pack("<II", 1, len(secs)) + b"".join(secs)
    vk = dict(alpha1=bytes(hdr1[0]), beta2=bytes(hdr2[0]), gamma2=bytes(hdr2[1]), delta2=bytes(hdr2[2]),
              ic=[bytes(pic[i]) for i in range(n_public + 1)])
    shape = dict(n_vars=n_vars, bit0=bit0, byte0=byte0, prod0=prod0, prods=prods)
    return dict(zkey=zkey, vk=vk, witness=witness_bytes(w), public=[w[1]], n_vars=n_vars, n_public=n_public, domain=N,
                n_coefs=len(coefs), shape=shape,
                new_witness=lambda wseed: (lambda ww: (witness_bytes(ww), [ww[1]]))(make_witness(wseed)))


def fast_witness(shape, wseed):
    """A satisfying assignment of the circuit `shape` (build()['shape']: picklable, so that every rank of a multi-GPU run can
    make its own witnesses) as ((n_vars, 32) uint8, [public input]): the free wires from numpy's generator (1.3 M Python
    big-int draws per witness are what make build()['new_witness'] take seconds), the product wires computed exactly."""
    n_vars, bit0, byte0, prod0 = shape["n_vars"], shape["bit0"], shape["byte0"], shape["prod0"]
    rs = np.random.RandomState(wseed & 0x7FFFFFFF)
    w = np.zeros((n_vars, 32), dtype=np.uint8)
    w[0, 0] = 1
    pub = (int.from_bytes(rs.bytes(8), "little") | 1)
    w[1, :8] = np.frombuffer(pub.to_bytes(8, "little"), dtype=np.uint8)
    w[bit0:byte0, 0] = rs.randint(0, 2, size=byte0 - bit0)
    w[byte0:prod0, 0] = rs.randint(0, 256, size=prod0 - byte0)
    val = {}

    def get(i):
        if i >= prod0:
            return val[i]
        return pub if i == 1 else int(w[i, 0])

    for c, a, b, d, k1, k2, k3 in shape["prods"]:
        v = (k1 * get(a) + k2 * get(b)) % R * (k3 * get(d) % R) % R
        val[c] = v
        w[c] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    return w, [pub]


def oracle_points(group, scalars):
    """scalar * G through the CPU oracle (small keys in CPU-only tests)."""
    import oracle_lib as ol
    g = ol.generator(group)
    out = np.zeros((len(scalars), ol.AFF_BYTES[group]), dtype=np.uint8)
    for i, k in enumerate(scalars):
        out[i] = np.frombuffer(ol.pt_to_affine(group, ol.mul_scalar(group, g, pm.limbs(k % R))), dtype=np.uint8)
    return out


def write_wtns(path, wit):
    sec1 = struct.pack("<I", 32) + pm.limbs(R) + struct.pack("<I", wit.shape[0])
    with open(path, "wb") as f:
        f.write(b"wtns" + struct.pack("<II", 2, 2) + _section(1, sec1) + _section(2, wit.tobytes()))
