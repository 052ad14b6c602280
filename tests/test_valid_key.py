"""A synthetic Groth16 key built from a known trapdoor (tests/valid_key_builder.py): the oracle's prove() on a
NON-toy circuit must produce proofs that verify -- the reference's acceptance criterion
(prover-service/src/tests/prover_handler.rs:279-290) instead of byte comparison -- under two independent verifiers:
the C restatement of ark-groth16 (oracle/pairing_ref.h) and the pure-Python pairing.  CPU only."""
import json

import bn254_pairing as bp
import groth16_io as gio
import oracle_lib as ol
import pymodel as pm
import valid_key_builder as vkb


def _snarkjs_vk(vk):
    g1 = lambda b: [str(v) for v in pm.g1_aff_from_bytes(b)] + ["1"]

    def g2(b):
        (xa, xb), (ya, yb) = pm.g2_aff_from_bytes(b)
        return [[str(xa), str(xb)], [str(ya), str(yb)], ["1", "0"]]
    return dict(vk_alpha_1=g1(vk["alpha1"]), vk_beta_2=g2(vk["beta2"]), vk_gamma_2=g2(vk["gamma2"]),
                vk_delta_2=g2(vk["delta2"]), IC=[g1(p) for p in vk["ic"]])


def test_oracle_proof_of_a_synthetic_circuit_verifies(tmp_path):
    key = vkb.build(vkb.oracle_points, n_bits=40, n_bytes=6, n_prod=9, seed=3)
    assert (key["n_vars"], key["n_public"], key["domain"]) == (57, 1, 64)
    zk, wt = str(tmp_path / "v.zkey"), str(tmp_path / "v.wtns")
    open(zk, "wb").write(key["zkey"])
    vkb.write_wtns(wt, key["witness"])
    assert ol.zkey_info(zk) == dict(n_vars=57, n_public=1, domain_size=64, n_coefs=key["n_coefs"])
    x = key["public"][0]
    for r, s in ((0, 0), (pm.SplitMix64(1).below(pm.R), pm.SplitMix64(2).below(pm.R))):
        js = ol.prove_files(zk, wt, pm.limbs(r), pm.limbs(s))
        proof = gio.proof_from_json(js)
        assert ol.groth16_verify(key["vk"], proof, [x])
        assert not ol.groth16_verify(key["vk"], proof, [x + 1])
    # the independent Python verifier agrees (accept and reject)
    assert bp.groth16_verify(_snarkjs_vk(key["vk"]), json.loads(js), [x])
    assert not bp.groth16_verify(_snarkjs_vk(key["vk"]), json.loads(js), [x + 2])
    # a witness that does not satisfy the circuit gives a proof that does not verify
    bad = key["witness"].copy()
    bad[5, 0] ^= 1                      # flip a bit wire that feeds nothing else: still a bit -> still satisfied
    bad[2 + 40 + 6, 0] ^= 1             # change a product wire: its defining constraint breaks
    vkb.write_wtns(wt, bad)
    js_bad = ol.prove_files(zk, wt, pm.limbs(1), pm.limbs(2))
    assert not ol.groth16_verify(key["vk"], gio.proof_from_json(js_bad), [x])
