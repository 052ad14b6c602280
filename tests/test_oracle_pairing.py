"""Pins oracle/pairing_ref.h (the CPU restatement of ark-ec 0.4.2's BN254 pairing and ark-groth16 0.4.0's verifier,
which the service runs on every proof: prover_handler.rs:329-336) -- CPU only.

  * the reference's own acceptance test: the toy circuit's proof verifies under toy_vk.json with public input 2
    (prover-service/src/tests/prover_handler.rs:279-290) and not with 3
  * an independent implementation: tests/bn254_pairing.py (pure Python, different tower basis, binary Miller loop,
    direct exponent (p^12-1)/r) raised to 2x(6x^2+3x+1), the factor of the Fuentes-Castaneda hard part
  * bilinearity, non-degeneracy, e(P, 0) = e(0, Q) = 1
"""
import json

import numpy as np

import bn254_pairing as bp
import groth16_io as gio
import oracle_lib as ol
import pymodel as pm
from test_oracle_prove import KNOWN_RS0

X = 4965661367192848881
LAMBDA = 2 * X * (6 * X * X + 3 * X + 1)


def gt_to_python_basis(gt):
    """Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - 9 - u)  ->  coefficients of 1, w, ..., w^11 with w^6 = 9 + u
    (the representation of tests/bn254_pairing.py): (x + y u) v^j w^k = x w^(2j+k) + y (w^6 - 9) w^(2j+k)."""
    c = [0] * 12
    vals = [pm.from_mont(pm.unlimbs(gt[32 * i:32 * i + 32]), pm.Q) for i in range(12)]
    for k in range(2):          # power of w (c0, c1)
        for j in range(3):      # power of v
            x, y = vals[(k * 3 + j) * 2], vals[(k * 3 + j) * 2 + 1]
            e = 2 * j + k
            c[e] = (c[e] + x - 9 * y) % pm.Q
            c[e + 6] = (c[e + 6] + y) % pm.Q
    return bp.F12(c)


def _g1(k):
    return pm.ec_mul(pm.Fq1Ops, (1, 2), k)


G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _g2(k):
    return pm.ec_mul(pm.Fq2Ops, G2_GEN, k)


def test_pairing_equals_python_pairing_to_the_fuentes_castaneda_factor():
    for a, b in ((1, 1), (5, 7)):
        P, Qp = _g1(a), _g2(b)
        gt = ol.pairing(pm.g1_aff_bytes(P), pm.g2_aff_bytes(Qp))
        want = bp.final_exp(bp.miller_loop(Qp, P)) ** LAMBDA
        assert gt_to_python_basis(gt) == want
        assert not (want == bp.F12.one())


def test_bilinearity_and_degenerate_inputs():
    P, Qp = pm.g1_aff_bytes(_g1(1)), pm.g2_aff_bytes(_g2(1))
    e = ol.pairing(P, Qp)
    e6 = e
    for _ in range(5):
        e6 = ol.gt_mul(e6, e)
    assert ol.pairing(pm.g1_aff_bytes(_g1(2)), pm.g2_aff_bytes(_g2(3))) == e6
    assert ol.pairing(pm.g1_aff_bytes(_g1(6)), Qp) == e6
    assert ol.pairing(P, pm.g2_aff_bytes(_g2(6))) == e6
    one = ol.pairing(bytes(64), Qp)
    assert one == ol.pairing(P, bytes(128))
    assert gt_to_python_basis(one) == bp.F12.one()
    # e(P, Q) * e(-P, Q) = 1 through the product of Miller loops and ONE final exponentiation (what the verifier does)
    negP = pm.g1_aff_bytes(pm.ec_neg(pm.Fq1Ops, _g1(1)))
    assert ol.final_exp(ol.gt_mul(ol.miller(P, Qp), ol.miller(negP, Qp))) == one


def test_toy_proof_accepts_with_2_rejects_with_3(toy_paths):
    zkey, wtns, vkp = toy_paths
    vk = gio.vk_from_json(vkp)
    assert len(vk["ic"]) == 2
    proof = gio.proof_from_json(KNOWN_RS0)                 # recorded from the unmodified reference prover
    assert ol.groth16_verify(vk, proof, [2])
    assert not ol.groth16_verify(vk, proof, [3])
    # seeded blinding, and agreement with the pure-Python verifier on accept and reject
    rng = pm.SplitMix64(0xC0FFEE)
    js = ol.prove_files(zkey, wtns, pm.limbs(rng.below(pm.R)), pm.limbs(rng.below(pm.R)))
    p2 = gio.proof_from_json(js)
    assert ol.groth16_verify(vk, p2, [2]) and bp.verify_json(vkp, js, [2])
    assert ol.groth16_verify(vk, p2, [2 + pm.R])            # inputs act modulo r (Fr::from_le_bytes_mod_order)
    # tampered proofs: A <-> C swapped, B replaced, a coordinate off the curve
    bad = p2[192:256] + p2[64:192] + p2[0:64]
    assert not ol.groth16_verify(vk, bad, [2])
    bad = p2[:64] + pm.g2_aff_bytes(_g2(5)) + p2[192:]
    assert not ol.groth16_verify(vk, bad, [2])
    d = json.loads(js)
    d["pi_c"][0] = str((int(d["pi_c"][0]) + 1) % pm.Q)
    assert not ol.groth16_verify(vk, gio.proof_from_json(d), [2])
    assert not bp.verify_json(vkp, json.dumps(d), [2])
