"""Full Groth16 proof on one MI355X at Keyless shape (BASELINE config 3) with a SYNTHETIC key:
nVars = 1,343,588, nPublic = 1, domainSize = 2^21, nCoefs ~ 8.3 M; the real Keyless zkey is
not available offline.  Reports proofs/s, p50 latency and the device-time split.  --check also runs the CPU
oracle once on the same files and compares the proof JSON byte for byte (minutes of CPU time).

    python tools/bench_proof.py [--scale 1.0] [--proofs 10] [--check]
"""
import argparse
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
import k16  # noqa: E402

# the regime the product runs in (bench.py, the harness): eight hardware queues, set before the
# process's first HIP call -- round 5's closing experiments ran this tool with the default four
k16.load().k16_runtime_hw_queues(int(os.environ.get("K16_TOOL_HW_QUEUES", "8")))

sys.path.insert(0, ROOT)
import bench as _bench  # noqa: E402  (the synthetic Keyless-shape key and witness are bench.py's)

R = _bench.R_MOD
le32 = _bench._le32
section = _bench._section


def synth_zkey_bytes(ctx, n_vars, n_public, N, n_coefs, seed=1):
    return _bench.synth_zkey_bytes(ctx, k16, n_vars, n_public, N, n_coefs, seed)


synth_witness = _bench.synth_witness


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the circuit (1.0 = Keyless shape)")
    ap.add_argument("--proofs", type=int, default=10)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--facade", type=int, default=0,
                    help="also drive the C++ FullProver facade (tests/cpp/fullprover_harness.cpp) with this many threads over "
                         "a pool of as many provers on GPU 0 (K16_DEVICES=0,0,..)")
    ap.add_argument("--no-stats", action="store_true", help="no per-stage HIP events inside the timed proofs")
    ap.add_argument("--random-rs", action="store_true", help="blinding scalars from the OS CSPRNG (the production path) instead of two fixed 64-bit values")
    ap.add_argument("--concurrent", type=int, default=1, help="throughput mode: this many provers (own context, streams) share the GPU")
    args = ap.parse_args()
    n_vars = max(int(1343588 * args.scale), 8)
    N = 1 << max(int(np.ceil(np.log2(max(1376867 * args.scale, 4)))), 2)
    n_coefs = int(8300000 * args.scale)
    ctx = k16.Context(0)
    t0 = time.time()
    zk = synth_zkey_bytes(ctx, n_vars, 1, N, n_coefs)
    print("synthetic zkey: nVars=%d N=2^%d nCoefs=%d  %.0f MB  (%.1f s)" % (n_vars, int(np.log2(N)), n_coefs, len(zk) / 1e6, time.time() - t0), flush=True)
    zpath = "/tmp/k16_synth.zkey"
    with open(zpath, "wb") as f:
        f.write(zk)
    del zk
    t0 = time.time()
    prover = k16.Prover(ctx, zpath)
    print("prover create (parse + CSR + upload + table prepare + roots): %.2f s" % (time.time() - t0), flush=True)
    r, s = le32(12345678901234567890 % R), le32(98765432109876543210 % R)
    wits = [synth_witness(n_vars, 100 + i) for i in range(min(args.proofs, 4))]
    prover.prove_mem(wits[0], r, s)  # warm-up: workspace allocation
    lat, dev = [], []
    ctx.stats_enable(not args.no_stats)
    ctx.stats_reset()
    t_all = time.perf_counter()
    for i in range(args.proofs):
        t1 = time.perf_counter()
        js = prover.prove_mem(wits[i % len(wits)]) if args.random_rs else prover.prove_mem(wits[i % len(wits)], r, s)
        lat.append((time.perf_counter() - t1) * 1e3)
        dev.append(prover.last_device_ms)
    total = time.perf_counter() - t_all
    stages = {k: ctx.stats_get(k)[1] / args.proofs for k in ("msm_sort", "msm_accumulate", "msm_fold", "msm_reduce", "ntt")}
    ctx.stats_enable(False)
    out = {"metric": "Groth16 proofs/s (synthetic Keyless-shape key, 1 MI355X)", "value": args.proofs / total,
           "p50_ms": float(np.median(lat)), "p99_ms": float(np.percentile(lat, 99)), "device_ms_p50": float(np.median(dev)),
           "stage_ms_per_proof": stages, "n_vars": n_vars, "domain": N, "n_coefs": n_coefs}
    print(json.dumps(out), flush=True)
    if args.concurrent > 1:
        # throughput mode: several provers on one GPU (e.g. one per service worker); the GPU interleaves their kernels
        import threading
        octx = [k16.Context(0) for _ in range(args.concurrent - 1)]
        if not os.environ.get("K16_BENCH_NO_SHARED_OPT"):
            for c in [ctx] + octx:
                c.set_option(k16.OPT_SHARED_GPU, 1)      # what FullProver does for K16_DEVICES=0,0
        provers = [prover] + [k16.Prover(c, zpath) for c in octx]
        for pv in provers:
            pv.prove_mem(wits[0], r, s)
        lats = [[] for _ in provers]

        def worker(i):
            for k in range(args.proofs):
                t1 = time.perf_counter()
                provers[i].prove_mem(wits[(i + k) % len(wits)], r, s)
                lats[i].append((time.perf_counter() - t1) * 1e3)
        t_all = time.perf_counter()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(provers))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        total = time.perf_counter() - t_all
        allv = [x for l in lats for x in l]
        print(json.dumps({"metric": "Groth16 proofs/s, throughput mode (%d provers sharing 1 MI355X)" % args.concurrent,
                          "value": args.proofs * len(provers) / total, "p50_ms": float(np.median(allv)),
                          "p99_ms": float(np.percentile(allv, 99))}), flush=True)
    if args.facade:
        import subprocess
        wpath = "/tmp/k16_synth.wtns"
        sec1 = struct.pack("<I", 32) + le32(R) + struct.pack("<I", n_vars)
        with open(wpath, "wb") as f:
            f.write(b"wtns" + struct.pack("<II", 2, 2) + section(1, sec1) + section(2, wits[0].tobytes()))
        exe = "/tmp/k16_fullprover_harness"
        pkg = os.path.join(ROOT, "keyless-zk-proofs_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "fullprover_harness.cpp"), "-L", pkg, "-lk16",
                               "-Wl,-rpath," + pkg, "-pthread", "-o", exe])
        env = dict(os.environ, K16_DEVICES=",".join(["0"] * args.facade))
        out = subprocess.run([exe, zpath, wpath, str(args.proofs), str(args.facade)], capture_output=True, text=True, env=env)
        last = [l for l in out.stdout.splitlines() if l.startswith("elapsed_ms=")]
        ok = sum(1 for l in out.stdout.splitlines() if l.startswith("type=0 error=0"))
        ms = float(last[0].split()[0].split("=")[1]) if last else float("nan")
        print(json.dumps({"metric": "Groth16 proofs/s through the C++ FullProver facade (%d threads, pool of %d provers on one MI355X)"
                                    % (args.facade, args.facade),
                          "value": ok / (ms * 1e-3), "proofs_ok": ok, "elapsed_ms": ms}), flush=True)
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        wpath = "/tmp/k16_synth.wtns"
        sec1 = struct.pack("<I", 32) + le32(R) + struct.pack("<I", n_vars)
        with open(wpath, "wb") as f:
            f.write(b"wtns" + struct.pack("<II", 2, 2) + section(1, sec1) + section(2, wits[0].tobytes()))
        t0 = time.time()
        want = ol.prove_files(zpath, wpath, r, s, nthreads=min(os.cpu_count() or 1, 16))
        cpu_s = time.time() - t0
        got = prover.prove_mem(wits[0], r, s)
        print(json.dumps({"check": "proof JSON equal to CPU oracle", "equal": got == want, "oracle_seconds": cpu_s}), flush=True)
        assert got == want
    prover.close()
    ctx.close()


if __name__ == "__main__":
    main()
