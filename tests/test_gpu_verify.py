"""-m gpu : the batched Groth16 verifier (k16_vk_create / k16_verify_batch / k16_pairing_vec, csrc/verify.hip) through the
C ABI, against the CPU oracle (oracle/pairing_ref.h, a restatement of ark-ec 0.4.2 / ark-groth16 0.4.0 pinned by
tests/test_oracle_pairing.py).  Replaces prover-service/src/request_handler/prover_handler.rs:329-336."""
import json

import numpy as np
import pytest

import groth16_io as gio
import oracle_lib as ol
import pymodel as pm
from test_oracle_prove import KNOWN_RS0

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import k16
    c = k16.Context(0)
    yield c
    c.close()


def test_pairing_values_bit_exact(ctx):
    """e(P, Q) as 12 x 32 bytes equal to the oracle's for multiples of the generators, zero points included (a zero P
    or Q gives 1, as ark-ec's multi_miller_loop skips the pair); 70 pairs = more than one wavefront."""
    import k16
    n = 70
    g1 = ol.gen_points(0, 10, n)
    g2 = ol.gen_points(1, 20, n)
    g1[4] = 0
    g2[5] = 0
    g1[6] = 0
    g2[6] = 0
    got = k16.pairing_vec(ctx, g1, g2)
    for i in range(n):
        assert got[i].tobytes() == ol.pairing(bytes(g1[i]), bytes(g2[i])), i
    # bilinearity on the device values: e(2P, 3Q) = e(P, Q)^6
    one = ol.gen_points(0, 0, 2)
    two = ol.gen_points(1, 0, 3)
    e = k16.pairing_vec(ctx, np.stack([one[0], one[1]]), np.stack([two[0], two[2]]))
    e6 = e[0].tobytes()
    for _ in range(5):
        e6 = ol.gt_mul(e6, e[0].tobytes())
    assert e[1].tobytes() == e6


def test_verify_toy_proofs_accept_reject(ctx, toy_paths):
    """The reference's acceptance criterion on the GPU: toy proofs verify under toy_vk.json with public input 2
    (prover-service/src/tests/prover_handler.rs:279-290), not with 3; tampered proofs are rejected; every flag equals
    the oracle's."""
    import k16
    zkey, wtns, vkp = toy_paths
    vk = gio.vk_from_json(vkp)
    V = k16.VerifyingKey(ctx, vk)
    known = gio.proof_from_json(KNOWN_RS0)
    assert V.verify_batch([known], [[2]]) == [True]
    assert V.verify_batch([known], [[3]]) == [False]
    # proofs made by the GPU prover with fresh CSPRNG blinding
    p = k16.Prover(ctx, zkey)
    fresh = [gio.proof_from_json(p.prove_file(wtns)) for _ in range(5)]
    p.close()
    g2x = ol.gen_points(1, 77, 1)[0].tobytes()
    d = json.loads(json.dumps(KNOWN_RS0))
    d["pi_c"][0] = str((int(d["pi_c"][0]) + 1) % pm.Q)          # off the curve
    cases = [(known, 2), (known, 3), (known, 2 + pm.R), (known, 0)]
    cases += [(f, 2) for f in fresh] + [(fresh[0], 1)]
    cases += [(known[192:256] + known[64:192] + known[0:64], 2)]  # A and C swapped
    cases += [(known[:64] + g2x + known[192:], 2)]                # B replaced
    cases += [(gio.proof_from_json(d), 2)]
    cases += [(bytes(64) + known[64:], 2), (known[:64] + bytes(128) + known[192:], 2), (bytes(256), 2)]   # zero points
    # a batch larger than a wavefront, accept and reject interleaved
    cases = cases * 6
    got = V.verify_batch([c[0] for c in cases], [[c[1]] for c in cases])
    want = [ol.groth16_verify(vk, c[0], [c[1]]) for c in cases]
    assert got == want
    assert got[0] and not got[1] and got[2] and all(got[4:9]) and not any(got[9:15])
    assert V.verify_batch([], []) == []
    V.close()


def test_verify_key_with_several_public_inputs(ctx):
    """n_ic > 2: vk_x = IC[0] + sum x_i IC[i+1] on the device.  A consistent (vk, proof) pair is built from known
    discrete logs: with A = aG, B = bH, C = cG, alpha = G, beta = H', gamma = H, delta = H the check reduces to
    ab = alpha*beta' + vkx + c (exponents), which fixes c."""
    import k16
    g = ol.generator(0)
    h = ol.generator(1)

    def g1(k):
        return ol.pt_to_affine(0, ol.mul_scalar(0, g, pm.limbs(k % pm.R)))

    def g2(k):
        return ol.pt_to_affine(1, ol.mul_scalar(1, h, pm.limbs(k % pm.R)))

    ic_k = [11, 22, 33, 44]
    xs = [5, pm.R - 3, 123456789]
    al, be, a, b = 7, 9, 1001, 2002
    vkx = ic_k[0] + sum(x * k for x, k in zip(xs, ic_k[1:]))
    c = (a * b - al * be - vkx) % pm.R
    vk = dict(alpha1=g1(al), beta2=g2(be), gamma2=g2(1), delta2=g2(1), ic=[g1(k) for k in ic_k])
    proof = g1(a) + g2(b) + g1(c)
    bad = g1(a) + g2(b) + g1(c + 1)
    assert ol.groth16_verify(vk, proof, xs) and not ol.groth16_verify(vk, bad, xs)
    V = k16.VerifyingKey(ctx, vk)
    assert V.verify_batch([proof, bad, proof], [xs, xs, [xs[0], xs[1], xs[2] + 1]]) == [True, False, False]
    V.close()


def _gpu_points(ctx):
    return lambda group, scalars: ctx.synth_points_scalars(group, scalars)


@pytest.mark.parametrize("shape", [(12000, 1000, 300), (1209229, 107487, 26870)], ids=["2p14", "keyless_shape"])
def test_proofs_of_a_valid_synthetic_key_verify(ctx, tmp_path, shape):
    """A Groth16 key built from a known trapdoor (tests/valid_key_builder.py; points scalar * G made on the device): the
    GPU prover's proofs must VERIFY -- the reference's acceptance criterion -- under the GPU verifier and under the oracle's,
    with the right public input and not with a wrong one; with injected (r, s) they are also byte-equal to the oracle's.
    The second shape is the Keyless circuit's: nVars 1,343,588, domain 2^21, 90 % bit / 8 % byte / 2 % full-width wires."""
    import os
    import k16
    import valid_key_builder as vkb
    key = vkb.build(_gpu_points(ctx), *shape, seed=7)
    if shape[0] > 100000:
        assert (key["n_vars"], key["domain"]) == (1343588, 1 << 21)
    zk, wt = str(tmp_path / "v.zkey"), str(tmp_path / "v.wtns")
    open(zk, "wb").write(key["zkey"])
    vkb.write_wtns(wt, key["witness"])
    x = key["public"][0]
    p = k16.Prover(ctx, zk)
    V = k16.VerifyingKey(ctx, key["vk"])
    proofs = [gio.proof_from_json(p.prove_mem(key["witness"])) for _ in range(3)]      # CSPRNG blinding
    assert len(set(proofs)) == 3
    assert V.verify_batch(proofs + proofs, [[x]] * 3 + [[x + 1]] * 3) == [True] * 3 + [False] * 3
    assert ol.groth16_verify(key["vk"], proofs[0], [x]) and not ol.groth16_verify(key["vk"], proofs[0], [x + 1])
    r, s = pm.limbs(pm.SplitMix64(91).below(pm.R)), pm.limbs(pm.SplitMix64(92).below(pm.R))
    got = p.prove_file(wt, r, s)
    assert got == ol.prove_files(zk, wt, r, s, nthreads=os.cpu_count() or 8)
    assert V.verify_batch([gio.proof_from_json(got)], [[x]]) == [True]
    # a witness that violates one constraint: the proof is produced, and rejected
    bad = key["witness"].copy()
    bad[key["n_vars"] - 1, 0] ^= 1
    assert V.verify_batch([gio.proof_from_json(p.prove_mem(bad))], [[x]]) == [False]
    V.close()
    p.close()


def test_config4_wave_of_distinct_witnesses_through_the_fullprover_pool(ctx, tmp_path):
    """BASELINE config 4 in the form one GPU allows, through the drop-in boundary: ONE FullProver whose pool holds three
    provers (K16_DEVICES=0,0,0; a node lists its eight GPUs), four concurrent callers, a wave of 16 DISTINCT witnesses of
    a VALID key of the Keyless shape (nVars 1,343,588, N = 2^21) read from .wtns files -- every proof must verify under
    ONE k16_verify_batch with its own public input, and be rejected with another proof's input / when tampered.
    Replaces the mutex around the single prover of prover-service/src/prover_state.rs:21."""
    import os
    import subprocess
    import k16
    import valid_key_builder as vkb
    from test_boundary import build_harness
    n_wit = 16
    key = vkb.build(_gpu_points(ctx), 1209229, 107487, 26870, seed=13)
    assert (key["n_vars"], key["domain"]) == (1343588, 1 << 21)
    zk = str(tmp_path / "wave.zkey")
    open(zk, "wb").write(key["zkey"])
    paths, inputs = [], []
    for i in range(n_wit):
        w, pub = key["new_witness"](500 + i)
        paths.append(str(tmp_path / ("w%02d.wtns" % i)))
        vkb.write_wtns(paths[-1], w)
        inputs.append(pub)
    assert len({tuple(p) for p in inputs}) == n_wit               # distinct public inputs, i.e. distinct statements
    exe = build_harness(tmp_path)
    env = dict(os.environ, K16_DEVICES="0,0,0")
    out = subprocess.run([exe, zk, ",".join(paths), "4", "4"], capture_output=True, text=True, timeout=900, env=env)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr[-2000:]
    proofs = [None] * n_wit
    for k in range(n_wit):
        head = lines[1 + 2 * k]
        assert head.startswith("type=0 error=0 ms="), lines[:6]
        proofs[int(head.split("wtns=")[1])] = gio.proof_from_json(lines[2 + 2 * k])
    assert all(p is not None for p in proofs) and len(set(proofs)) == n_wit
    V = k16.VerifyingKey(ctx, key["vk"])
    tampered = bytearray(proofs[3])
    tampered[200] ^= 1                                           # one bit of C.x
    batch = proofs + [proofs[0], bytes(tampered)]
    ins = inputs + [inputs[1], inputs[3]]                         # proof 0 with witness 1's public input; tampered proof 3
    assert V.verify_batch(batch, ins) == [True] * n_wit + [False, False]
    assert ol.groth16_verify(key["vk"], proofs[5], inputs[5])     # and the CPU oracle agrees on one of them
    V.close()


@pytest.fixture(scope="module")
def keyless_valid_key(ctx, tmp_path_factory):
    """ONE valid synthetic key of the Keyless shape (18 s of host big-integer work) for the round-6 pool / wave tests."""
    import valid_key_builder as vkb
    key = vkb.build(_gpu_points(ctx), 1209229, 107487, 26870, seed=17)
    assert (key["n_vars"], key["domain"]) == (1343588, 1 << 21)
    zk = str(tmp_path_factory.mktemp("valid_key") / "keyless_valid.zkey")
    open(zk, "wb").write(key["zkey"])
    key["zkey"] = None          # (0.9 GB: the file is what the provers read)
    return key, zk


def test_provers_sharing_one_resident_key(ctx, keyless_valid_key):
    """k16_prover_create_shared (include/k16.h): a second prover of the same key on the same device shares
    the first one's read-only device data by reference count instead of uploading and preparing 2.7 GB again.  At the Keyless
    shape: created in a fraction of the first one's time; its proofs are byte-equal to the first prover's (injected (r, s))
    and verify; both prove concurrently; the shared part outlives the prover that uploaded it.
    Replaces the per-instance zkey mapping of RS/fullprover.cpp:136-181."""
    import threading
    import time
    import k16
    key, zk = keyless_valid_key
    c1, c2, c3 = k16.Context(0), k16.Context(0), k16.Context(0)
    try:
        t0 = time.perf_counter()
        p1 = k16.Prover(c1, zk)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        p2 = k16.Prover(c2, zk, share_key_of=p1)
        t_shared = time.perf_counter() - t0
        assert p2.info() == p1.info()
        assert t_shared < 0.5 * t_first and t_shared < 0.6, (t_first, t_shared)     # (includes its warm-up proof)
        with pytest.raises(k16.K16Error):
            k16.Prover(c1, zk, share_key_of=p1)                                      # needs a context of its own
        V = k16.VerifyingKey(ctx, key["vk"])
        import valid_key_builder as vkb
        w, pub = vkb.fast_witness(key["shape"], 901)
        r, s = pm.limbs(pm.SplitMix64(71).below(pm.R)), pm.limbs(pm.SplitMix64(72).below(pm.R))
        a = p1.prove_mem(w, r, s)
        assert p2.prove_mem(w, r, s) == a
        assert V.verify_batch([gio.proof_from_json(a)], [pub]) == [True]
        # both at once, distinct witnesses, CSPRNG blinding: every proof verifies with ITS input
        wits = [vkb.fast_witness(key["shape"], 910 + i) for i in range(4)]
        out = [[None] * 4, [None] * 4]

        def work(k, pv):
            for i in range(4):
                out[k][i] = pv.prove_mem(wits[(i + k) % 4][0])

        th = [threading.Thread(target=work, args=(k, pv)) for k, pv in enumerate((p1, p2))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        proofs = [gio.proof_from_json(out[k][i]) for k in range(2) for i in range(4)]
        ins = [wits[(i + k) % 4][1] for k in range(2) for i in range(4)]
        assert V.verify_batch(proofs, ins) == [True] * 8
        # the prover that uploaded the key goes first: the shared part stays until its last user has gone
        p1.close()
        c1.close()
        assert p2.prove_mem(w, r, s) == a
        p3 = k16.Prover(c3, zk, share_key_of=p2)
        p2.close()
        assert p3.prove_mem(w, r, s) == a
        p3.close()
        V.close()
    finally:
        for c in (c3, c2):
            c.close()


def test_fullprover_pool_compact_hand_off_and_shared_key(ctx, tmp_path, keyless_valid_key):
    """The compact witness hand-off THROUGH THE POOL (k16_fullprover_compact_lease / _prove_compact, include/k16.h) on ONE
    FullProver whose two provers share device 0 and one resident key (K16_DEVICES=0,0): three concurrent callers lease a slot,
    write their witness into ITS pinned buffers and prove on it; every proof of the valid Keyless-shape key verifies with its
    own public input.  Replaces the temp-file hand-off of RS/fullprover.cpp:204-250 / prover_handler.rs:511-527."""
    import os
    import subprocess
    import k16
    import valid_key_builder as vkb
    from test_boundary import build_harness
    key, zk = keyless_valid_key
    n_wit = 6
    paths, inputs = [], []
    for i in range(n_wit):
        w, pub = vkb.fast_witness(key["shape"], 700 + i)
        paths.append(str(tmp_path / ("c%02d.wtns" % i)))
        vkb.write_wtns(paths[-1], w)
        inputs.append(pub)
    exe = build_harness(tmp_path)
    env = dict(os.environ, K16_DEVICES="0,0", K16_HARNESS_MEM="compact")
    out = subprocess.run([exe, zk, ",".join(paths), "2", "3"], capture_output=True, text=True, timeout=900, env=env)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr[-2000:]
    proofs = [None] * n_wit
    us = []
    for k in range(n_wit):
        head = lines[1 + 2 * k]
        assert head.startswith("type=0 error=0 ms="), lines[:8]
        us.append(int(head.split("ms=")[1].split()[0]))          # (compact mode: microseconds of the prove call alone)
        proofs[int(head.split("wtns=")[1])] = gio.proof_from_json(lines[2 + 2 * k])
    assert all(p is not None for p in proofs) and len(set(proofs)) == n_wit
    V = k16.VerifyingKey(ctx, key["vk"])
    assert V.verify_batch(proofs + [proofs[0]], inputs + [inputs[1]]) == [True] * n_wit + [False]
    V.close()
    assert 1000 < min(us) < 60000, us


def test_config4_wave_of_64_distinct_proofs_one_verification_batch(ctx, keyless_valid_key):
    """BASELINE config 4 at its stated wave size on whatever the box has: 64 DISTINCT witnesses of a valid Keyless-shape key
    (nVars 1,343,588, N = 2^21), one prover per visible GPU (two sharing device 0 and one resident key on a one-GPU box),
    every proof accepted by ONE k16_verify_batch with its own public input, and rejected with its neighbour's.
    The service's loop: prover_handler.rs:244-345."""
    import threading
    import k16
    import valid_key_builder as vkb
    key, zk = keyless_valid_key
    wits = [vkb.fast_witness(key["shape"], 3000 + i) for i in range(64)]
    assert len({tuple(w[1]) for w in wits}) == 64
    ndev = max(1, k16.load().k16_device_count())
    devs = list(range(ndev)) if ndev > 1 else [0, 0]
    ctxs = [k16.Context(d) for d in devs]
    provers = []
    for c, d in zip(ctxs, devs):
        sib = next((pv for pv, dd in zip(provers, devs) if dd == d), None)
        provers.append(k16.Prover(c, zk, share_key_of=sib))
    jobs, out, lock = list(range(64)), [None] * 64, threading.Lock()

    def worker(pv):
        while True:
            with lock:
                if not jobs:
                    return
                j = jobs.pop(0)
            out[j] = pv.prove_mem(wits[j][0])

    th = [threading.Thread(target=worker, args=(pv,)) for pv in provers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    proofs = [gio.proof_from_json(js) for js in out]
    assert len(set(proofs)) == 64
    V = k16.VerifyingKey(ctx, key["vk"])
    assert V.verify_batch(proofs, [w[1] for w in wits]) == [True] * 64
    assert V.verify_batch(proofs, [wits[(j + 1) % 64][1] for j in range(64)]) == [False] * 64
    assert ol.groth16_verify(key["vk"], proofs[37], wits[37][1])
    V.close()
    for pv in provers:
        pv.close()
    for c in ctxs:
        c.close()


def _neg_g2(q):
    """-(x, y) for an affine Montgomery G2 point (Montgomery form is linear: -yR = p - yR)."""
    q = bytes(q)
    out = bytearray(q[:64])
    for k in (64, 96):
        v = int.from_bytes(q[k:k + 32], "little")
        out += ((pm.Q - v) % pm.Q).to_bytes(32, "little")
    return bytes(out)


def _check_gt(vk, proof, xs):
    """e(A,B) e(vk_x,-gamma) e(C,-delta) from the ORACLE's pairing and group operations."""
    acc = None
    for j, x in enumerate(xs):
        t = ol.mul_scalar(0, vk["ic"][j + 1], int(x % (1 << 256)).to_bytes(32, "little"))
        acc = t if acc is None else ol.pt_op(0, 0, acc, t)
    vkx = ol.pt_to_affine(0, ol.pt_op(0, 1, acc, vk["ic"][0]))
    g = ol.pairing(proof[:64], proof[64:192])
    g = ol.gt_mul(g, ol.pairing(vkx, _neg_g2(vk["gamma2"])))
    return ol.gt_mul(g, ol.pairing(proof[192:256], _neg_g2(vk["delta2"])))


def test_wave_cooperative_verifier_gt_values_and_flags(ctx, toy_paths, monkeypatch):
    """The latency path of k16_verify_batch (one wavefront per proof, csrc/verify_script.h): the GT value it computes is
    byte-equal to the oracle's e(A,B) e(vk_x,-gamma) e(C,-delta) for accepted AND rejected proofs, its flags equal the
    general (one lane per pairing) path's and the oracle's, and inputs it does not cover (a zero point) still get the
    right flags through the general path.  prover_handler.rs:329-336 is the per-proof check this serves."""
    import k16
    zkey, wtns, vkp = toy_paths
    vk = gio.vk_from_json(vkp)
    V = k16.VerifyingKey(ctx, vk)
    known = gio.proof_from_json(KNOWN_RS0)
    p = k16.Prover(ctx, zkey)
    fresh = [gio.proof_from_json(p.prove_file(wtns)) for _ in range(3)]
    p.close()
    tampered = known[:64] + ol.gen_points(1, 77, 1)[0].tobytes() + known[192:]
    cases = [(known, 2), (known, 3), (known, 2 + pm.R), (known, 0), (fresh[0], 2), (fresh[1], 2), (fresh[2], 7), (tampered, 2),
             (known[192:256] + known[64:192] + known[0:64], 2), (known, (1 << 256) - 1)]
    gts = V.coop_gt([c[0] for c in cases], [[c[1]] for c in cases])
    for (pr, x), got in zip(cases, gts):
        assert got.tobytes() == _check_gt(vk, pr, [x]), x
    eab = ol.pairing(vk["alpha1"], vk["beta2"])
    want = [ol.groth16_verify(vk, c[0], [c[1]]) for c in cases]
    assert [g.tobytes() == eab for g in gts] == want
    assert V.verify_batch([c[0] for c in cases], [[c[1]] for c in cases]) == want          # cooperative path
    assert want[0] and not want[1] and want[2] and want[4] and want[5] and not want[6] and not want[7]
    # one proof alone (the service's case), and a batch larger than the number of CUs
    assert V.verify_batch([known], [[2]]) == [True] and V.verify_batch([known], [[3]]) == [False]
    big = cases * 60
    assert V.verify_batch([c[0] for c in big], [[c[1]] for c in big]) == want * 60
    # a zero point sends the batch to the general path: same flags as before
    withzero = cases + [(bytes(64) + known[64:], 2)]
    assert V.verify_batch([c[0] for c in withzero], [[c[1]] for c in withzero]) == want + [ol.groth16_verify(vk, withzero[-1][0], [2])]
    V.close()
    # the general path alone (cooperative path switched off at key creation) gives the same flags
    monkeypatch.setenv("K16_VERIFY_NO_COOP", "1")
    ctx2 = k16.Context(0)            # the switch is read when a context is created
    try:
        V2 = k16.VerifyingKey(ctx2, vk)
        assert V2.verify_batch([c[0] for c in cases], [[c[1]] for c in cases]) == want
        with pytest.raises(k16.K16Error):
            V2.coop_gt([known], [[2]])
        V2.close()
    finally:
        ctx2.close()


def test_wave_cooperative_verifier_several_public_inputs(ctx):
    """n_ic = 4: the window tables of IC[1..3] and the lane-per-window sum of vk_x."""
    import k16
    g, h = ol.generator(0), ol.generator(1)

    def g1(k):
        return ol.pt_to_affine(0, ol.mul_scalar(0, g, pm.limbs(k % pm.R)))

    def g2(k):
        return ol.pt_to_affine(1, ol.mul_scalar(1, h, pm.limbs(k % pm.R)))

    ic_k = [11, 22, 33, 44]
    xs = [5, pm.R - 3, 123456789]
    al, be, a, b = 7, 9, 1001, 2002
    vkx = ic_k[0] + sum(x * k for x, k in zip(xs, ic_k[1:]))
    c = (a * b - al * be - vkx) % pm.R
    vk = dict(alpha1=g1(al), beta2=g2(be), gamma2=g2(1), delta2=g2(1), ic=[g1(k) for k in ic_k])
    proof, bad = g1(a) + g2(b) + g1(c), g1(a) + g2(b) + g1(c + 1)
    V = k16.VerifyingKey(ctx, vk)
    gts = V.coop_gt([proof, bad], [xs, xs])
    assert gts[0].tobytes() == _check_gt(vk, proof, xs) == ol.pairing(vk["alpha1"], vk["beta2"])
    assert gts[1].tobytes() == _check_gt(vk, bad, xs)
    assert V.verify_batch([proof, bad, proof], [xs, xs, [xs[0], xs[1], xs[2] + 1]]) == [True, False, False]
    # vk_x at infinity (x chosen so that IC[0] + x IC[1] = O with the other inputs zero): left to the general path
    x0 = (-ic_k[0] * pow(ic_k[1], -1, pm.R)) % pm.R
    assert V.verify_batch([proof], [[x0, 0, 0]]) == [ol.groth16_verify(vk, proof, [x0, 0, 0])]
    V.close()


def test_verifier_rejects_non_canonical_and_off_curve_points(ctx, toy_paths, monkeypatch):
    """ark's deserialisation refuses non-canonical encodings and points off the curve before verify_proof sees them (round-2
    advisor finding: A, A + p and A + 2p all fit 256 bits and verified identically, so proof bytes were malleable).  Both GPU
    paths now reject them; the valid proof is still accepted."""
    import k16
    zkey, wtns, vkp = toy_paths
    vk = gio.vk_from_json(vkp)
    known = gio.proof_from_json(KNOWN_RS0)

    def bump(proof, off, delta):
        v = int.from_bytes(proof[off:off + 32], "little") + delta
        return proof[:off] + (v % (1 << 256)).to_bytes(32, "little") + proof[off + 32:]

    bad = [bump(known, 0, pm.Q), bump(known, 32, pm.Q), bump(known, 64, pm.Q), bump(known, 160, pm.Q), bump(known, 192, pm.Q),
           bump(known, 0, 2 * pm.Q),           # A.x + p, A.y + p, B.x.a + p, B.y.b + p, C.x + p, A.x + 2p: same residues
           bump(known, 32, 1), bump(known, 96, 1), bump(known, 224, 1)]   # off the curve / the twist
    for no_coop in (False, True):
        cx = ctx
        if no_coop:
            monkeypatch.setenv("K16_VERIFY_NO_COOP", "1")
            cx = k16.Context(0)      # the switch is read when a context is created
        V = k16.VerifyingKey(cx, vk)
        assert V.verify_batch([known] + bad, [[2]] * (1 + len(bad))) == [True] + [False] * len(bad), no_coop
        assert V.verify_batch([bad[0]], [[2]]) == [False]
        V.close()
        if no_coop:
            cx.close()
