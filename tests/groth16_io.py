"""snarkjs JSON (verification_key.json, proof.json) <-> the byte formats of the C ABIs: affine points in Montgomery
form, little-endian, G1 = x | y (64 B), G2 = x.a | x.b | y.a | y.b (128 B) -- the zkey's own point format (SURVEY
Appendix A).  The reference does the same conversion in Rust (prover-service/src/request_handler/types.rs:141-214:
prepared_vk / encode_proof build arkworks points from the decimal strings)."""
import json

import pymodel as pm


def g1(v):
    return pm.g1_aff_bytes((int(v[0]), int(v[1])))


def g2(v):
    return pm.g2_aff_bytes(((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1]))))


def vk_from_json(path_or_dict):
    vk = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
    return dict(alpha1=g1(vk["vk_alpha_1"]), beta2=g2(vk["vk_beta_2"]), gamma2=g2(vk["vk_gamma_2"]),
                delta2=g2(vk["vk_delta_2"]), ic=[g1(p) for p in vk["IC"]])


def proof_from_json(js):
    p = js if isinstance(js, dict) else json.loads(js)
    return g1(p["pi_a"]) + g2(p["pi_b"]) + g1(p["pi_c"])
