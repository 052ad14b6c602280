"""Times the batched GPU verifier (k16_verify_batch) on toy-circuit proofs next to the CPU oracle's verification.
    python tools/bench_verify.py [--batches 1,3,64,512,4096]
The cost of a Groth16 verification does not depend on the circuit (3 pairings + one scalar multiplication per public
input), so the toy key -- the only one with a verification key offline -- measures the Keyless case too (1 public input)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "keyless-zk-proofs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import k16  # noqa: E402
import groth16_io as gio  # noqa: E402
import oracle_lib as ol  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,3,64,512,4096,16384")
    args = ap.parse_args()
    toy = os.path.join(ROOT, "tests", "golden", "toy")
    ctx = k16.Context(0)
    vk = gio.vk_from_json(os.path.join(toy, "toy_vk.json"))
    t0 = time.perf_counter()
    V = k16.VerifyingKey(ctx, vk)
    t_vk = (time.perf_counter() - t0) * 1e3
    p = k16.Prover(ctx, os.path.join(toy, "toy_1.zkey"))
    proofs = [gio.proof_from_json(p.prove_file(os.path.join(toy, "toy.wtns"))) for _ in range(16)]
    p.close()
    t0 = time.perf_counter()
    for pr in proofs[:8]:
        assert ol.groth16_verify(vk, pr, [2])
    cpu_ms = (time.perf_counter() - t0) / 8 * 1e3
    out = {"vk_create_ms": t_vk, "cpu_oracle_ms_per_proof": cpu_ms, "batches": []}
    for n in [int(x) for x in args.batches.split(",")]:
        pr = [proofs[i % 16] for i in range(n)]
        inp = [[2 if i % 5 else 3] for i in range(n)]
        V.verify_batch(pr, inp)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ok = V.verify_batch(pr, inp)
        ms = (time.perf_counter() - t0) / reps * 1e3
        assert ok == [bool(i % 5) for i in range(n)]
        out["batches"].append({"n": n, "ms": ms, "proofs_per_s": n / ms * 1e3})
    print(json.dumps(out))
    V.close()
    ctx.close()


if __name__ == "__main__":
    main()
