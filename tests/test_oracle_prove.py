"""Pins the oracle's full prove() pipeline (zkey/wtns parse -> MSMs -> NTT chain -> blinding -> JSON)
on the only Groth16 fixture the reference ships: prover-service/resources/toy_circuit."""
import json

import bn254_pairing as bp
import oracle_lib as ol
import pymodel as pm

# Known answer with r = s = 0, produced by the reference's own FullProver on toy_1.zkey / toy.wtns
# (recorded in SURVEY.md section 8(c) from a run of the unmodified reference sources).
KNOWN_RS0 = {
    "pi_a": ["15497094993276509239765276704604591677364093863113574503272626306469959942084",
             "6385825892212757609423427245093616574110698284563433009382134021422919787741", "1"],
    "pi_b": [["10823290944885887106245988328395906657376942190842390555967597542424554222129",
              "19520083315704656987706813084942699153630269200025037381002848804737256842439"],
             ["6916087550021968361945280586838467978933260702187883004039489080438038746321",
              "5436277912204670347762553019758322958377559984844731918484666101590722971723"],
             ["1", "0"]],
    "pi_c": ["15688317691885337868523383412127187629833673227210016285038656060032986357197",
             "890484067630937953778630633794880096460812969865242013888153800710356792892", "1"],
    "protocol": "groth16",
}


def test_toy_header(toy_paths):
    zkey, _, _ = toy_paths
    assert ol.zkey_info(zkey) == dict(n_vars=3, n_public=1, domain_size=4, n_coefs=4)


def test_toy_known_answer_rs0(toy_paths):
    zkey, wtns, _ = toy_paths
    js = ol.prove_files(zkey, wtns)
    assert json.loads(js) == KNOWN_RS0
    # compact nlohmann dump(): sorted keys, no whitespace (groth16.cpp:378-410, fullprover.cpp:246)
    assert js == json.dumps(KNOWN_RS0, separators=(",", ":"), sort_keys=True)


def test_toy_verifies_with_public_input_2(toy_paths):
    """The reference's acceptance criterion (tests/prover_handler.rs:279-290): the toy proof
    verifies under toy_vk.json with public input 2 -- for r = s = 0 and for seeded blinding."""
    zkey, wtns, vk = toy_paths
    assert bp.verify_json(vk, ol.prove_files(zkey, wtns), [2])
    rng = pm.SplitMix64(0xBADC0DE)
    r, s = rng.below(pm.R), rng.below(pm.R)
    js = ol.prove_files(zkey, wtns, pm.limbs(r), pm.limbs(s))
    assert json.loads(js) != KNOWN_RS0
    assert bp.verify_json(vk, js, [2])
    assert not bp.verify_json(vk, js, [3])


def test_threads_do_not_change_the_proof(toy_paths):
    zkey, wtns, _ = toy_paths
    r, s = pm.limbs(12345), pm.limbs(pm.R - 1)
    assert ol.prove_files(zkey, wtns, r, s, nthreads=1) == ol.prove_files(zkey, wtns, r, s, nthreads=4)


def test_synthetic_circuit_h_scalars_against_python_model(tmp_path):
    """Pins the SpMV -> a*b -> 3 x (iNTT, coset shift, NTT) -> a*b - c chain (groth16.cpp:116-275) of the
    oracle on a small random circuit against an independent big-int evaluation:
    h[i] = (A.w)(x_i) * (B.w)(x_i) - ((A.w)*(B.w) on the domain)(x_i) at the coset points x_i = g_{2N}^(2i+1)."""
    import numpy as np
    import zkey_builder as zb
    n_vars, n_pub, N, n_coefs = 40, 2, 16, 70
    zk, wt = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    zb.build_zkey(zk, n_vars, n_pub, N, n_coefs, seed=3)
    w = zb.build_wtns(wt, n_vars, seed=4)
    _, h = ol.prove_files(zk, wt, want_h=True)
    got = [sum(int(h[i, k]) << (64 * k) for k in range(4)) for i in range(N)]
    # model
    import struct
    data = open(zk, "rb").read()
    # locate section 4
    pos, secs = 12, {}
    for _ in range(struct.unpack_from("<I", data, 8)[0]):
        t, sz = struct.unpack_from("<IQ", data, pos)
        secs[t] = (pos + 12, sz)
        pos += 12 + sz
    o, sz = secs[4]
    nc = struct.unpack_from("<I", data, o)[0]
    wv = [pm.unlimbs(bytes(w[i])) for i in range(n_vars)]
    a, b = [0] * N, [0] * N
    rinv2 = pow(pm.MONT, -2, pm.R)
    for i in range(nc):
        m, c, s = struct.unpack_from("<III", data, o + 4 + 44 * i)
        val = pm.unlimbs(data[o + 4 + 44 * i + 12:o + 4 + 44 * i + 44]) * rinv2 % pm.R
        (a if m == 0 else b)[c] = ((a if m == 0 else b)[c] + wv[s] * val) % pm.R
    cc = [x * y % pm.R for x, y in zip(a, b)]
    lg = 4
    g2n = pm.root_of_unity(lg + 1)

    def to_coset(ev):
        co = pm.intt_naive(ev, lg)
        co = [v * pow(g2n, i, pm.R) % pm.R for i, v in enumerate(co)]
        return pm.ntt_naive(co, lg)
    A, B, C = to_coset(a), to_coset(b), to_coset(cc)
    want = [(x * y - z) % pm.R for x, y, z in zip(A, B, C)]
    assert got == want


def test_threads_do_not_change_proof_or_h_at_parallel_sizes(tmp_path):
    """N >= 8192 switches the oracle's butterfly / pointwise loops and its MSM slices to OpenMP (the reference runs the
    same loops under tbb::parallel_for): proof JSON and H scalars must not depend on the thread count."""
    import numpy as np
    import zkey_builder as zb
    zk, wt = str(tmp_path / "s.zkey"), str(tmp_path / "s.wtns")
    zb.build_zkey(zk, 20000, 1, 1 << 15, 60000, seed=3)
    zb.build_wtns(wt, 20000, seed=4)
    r, s = pm.limbs(5), pm.limbs(7)
    a, h1 = ol.prove_files(zk, wt, r, s, nthreads=1, want_h=True)
    b, h2 = ol.prove_files(zk, wt, r, s, nthreads=8, want_h=True)
    assert a == b and np.array_equal(h1, h2)
