"""BASELINE config 4 in the form one GPU allows: a wave of 64 distinct Keyless-shape proofs from a pool of provers sharing
the GPU (one process, one context + resident key per prover, as K16_DEVICES=0,0,0 behind the FullProver facade; a node runs
the same pool with one or more provers per GPU), then ONE batched GPU verification of the whole wave -- the check the
service makes per proof before it releases it (prover_handler.rs:329-336).  The key is a VALID synthetic key of the Keyless
shape (tests/valid_key_builder.py), so the proofs really verify.

    python tools/config4_wave.py [--provers 3] [--wave 64] [--witnesses 8]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "keyless-zk-proofs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import k16  # noqa: E402
import groth16_io as gio  # noqa: E402
import valid_key_builder as vkb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--provers", type=int, default=3)
    ap.add_argument("--wave", type=int, default=64)
    ap.add_argument("--witnesses", type=int, default=64)
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()
    ctx = k16.Context(0)
    t0 = time.time()
    key = vkb.build(lambda g, sc: ctx.synth_points_scalars(g, sc), int(1209229 * args.scale), int(107487 * args.scale),
                    int(26870 * args.scale), seed=11)
    zpath = "/tmp/k16_config4_%d.zkey" % os.getpid()
    open(zpath, "wb").write(key["zkey"])
    t_key = time.time() - t0
    wits = [vkb.fast_witness(key["shape"], 100 + i) for i in range(args.witnesses)]
    ctxs = [ctx] + [k16.Context(0) for _ in range(args.provers - 1)]
    provers = [k16.Prover(ctx, zpath)]
    provers += [k16.Prover(c, zpath, share_key_of=provers[0]) for c in ctxs[1:]]     # one resident key (round 6)
    V = k16.VerifyingKey(ctx, key["vk"])
    for pv in provers:
        pv.prove_mem(wits[0][0])
    jobs = list(range(args.wave))
    out, lat, lock = [None] * args.wave, [0.0] * args.wave, threading.Lock()

    def worker(pv):
        while True:
            with lock:
                if not jobs:
                    return
                j = jobs.pop(0)
            t1 = time.perf_counter()
            out[j] = pv.prove_mem(wits[j % len(wits)][0])
            lat[j] = (time.perf_counter() - t1) * 1e3

    t_all = time.perf_counter()
    th = [threading.Thread(target=worker, args=(pv,)) for pv in provers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_prove = time.perf_counter() - t_all
    proofs = [gio.proof_from_json(js) for js in out]
    inputs = [wits[j % len(wits)][1] for j in range(args.wave)]
    V.verify_batch(proofs[:2], inputs[:2])                        # warm-up of the verifier kernels
    t1 = time.perf_counter()
    ok = V.verify_batch(proofs, inputs)
    t_verify = time.perf_counter() - t1
    wrong = V.verify_batch(proofs[:8], [[x[0] + 1] for x in inputs[:8]])
    res = {"config": "BASELINE config 4 on ONE MI355X: wave of %d Keyless-shape proofs, %d provers sharing the GPU, "
                     "one batched GPU verification" % (args.wave, args.provers),
           "n_vars": key["n_vars"], "domain": key["domain"], "n_coefs": key["n_coefs"], "distinct_witnesses": len(wits),
           "key": "valid synthetic key from a known trapdoor (proofs verify); built in %.1f s" % t_key,
           "prove_wave_s": t_prove, "proofs_per_s_proving": args.wave / t_prove,
           "prove_latency_p50_ms": float(np.median(lat)), "prove_latency_p99_ms": float(np.percentile(lat, 99)),
           "verify_wave_ms": t_verify * 1e3, "all_verified": bool(all(ok)), "wrong_input_rejected": not any(wrong),
           "proofs_per_s_proved_and_verified": args.wave / (t_prove + t_verify)}
    print(json.dumps(res))
    assert all(ok) and not any(wrong)
    os.unlink(zpath)


if __name__ == "__main__":
    main()
