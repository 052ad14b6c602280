"""Robustness fuzz of the file boundary: mutated .zkey / .wtns files (bit flips, truncations, section-length and header
edits of the reference's toy fixture and of a synthetic circuit) through k16_prover_create / k16_prover_prove_file.  Any
outcome is acceptable -- an error code, or a proof -- except a crash or a hang: each batch runs in a child process, a
dead child is the finding.  python tools/format_fuzz.py [cases] [seed]"""
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "keyless-zk-proofs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def mutate(rs, data, what):
    b = bytearray(data)
    k = rs.randint(6)
    if k == 0:                                   # a few random byte flips anywhere
        for _ in range(rs.randint(1, 6)):
            b[rs.randint(len(b))] ^= 1 << rs.randint(8)
    elif k == 1:                                 # truncation
        b = b[:rs.randint(0, len(b))]
    elif k == 2:                                 # flips inside the first 200 bytes (magic, version, section table, header)
        for _ in range(rs.randint(1, 4)):
            b[rs.randint(min(200, len(b)))] ^= 1 << rs.randint(8)
    elif k == 3:                                 # a section length replaced by something wild
        pos, nsec = 12, struct.unpack_from("<I", b, 8)[0] if len(b) >= 12 else 0
        secs = []
        for _ in range(min(nsec, 32)):
            if pos + 12 > len(b):
                break
            secs.append(pos)
            pos += 12 + struct.unpack_from("<Q", b, pos + 4)[0]
        if secs:
            at = secs[rs.randint(len(secs))]
            struct.pack_into("<Q", b, at + 4, int(rs.choice([0, 1, 31, 2 ** 31, 2 ** 40, 2 ** 63, len(b), len(b) * 2])))
    elif k == 4:                                 # header integers (n8, nVars, nPublic, domain) set to edge values
        for _ in range(rs.randint(1, 3)):
            at = rs.randint(0, max(1, min(len(b) - 4, 400)))
            struct.pack_into("<I", b, at, int(rs.choice([0, 1, 2, 3, 31, 33, 2 ** 16, 2 ** 31 - 1, 2 ** 32 - 1])))
    else:                                        # appended garbage / duplicated tail
        b += bytes(rs.randint(0, 256, size=rs.randint(1, 64)).astype(np.uint8))
    return bytes(b)


def child(cases, seed):
    import k16
    import zkey_builder as zb
    rs = np.random.RandomState(seed)
    ctx = k16.Context(0)
    d = tempfile.mkdtemp()
    toy_z = open(os.path.join(ROOT, "tests", "golden", "toy", "toy_1.zkey"), "rb").read()
    toy_w = open(os.path.join(ROOT, "tests", "golden", "toy", "toy.wtns"), "rb").read()
    zb.build_zkey(d + "/s.zkey", 300, 2, 64, 500, seed=5)
    zb.build_wtns(d + "/s.wtns", 300, seed=6)
    syn_z, syn_w = open(d + "/s.zkey", "rb").read(), open(d + "/s.wtns", "rb").read()
    stats = {"create_err": 0, "prove_err": 0, "proofs": 0}
    for c in range(cases):
        z, w = (toy_z, toy_w) if rs.rand() < 0.5 else (syn_z, syn_w)
        which = rs.randint(3)                    # mutate the key, the witness, or both
        zk = mutate(rs, z, "z") if which != 1 else z
        wt = mutate(rs, w, "w") if which != 0 else w
        open(d + "/m.zkey", "wb").write(zk)
        open(d + "/m.wtns", "wb").write(wt)
        try:
            p = k16.Prover(ctx, d + "/m.zkey")
        except k16.K16Error:
            stats["create_err"] += 1
            continue
        try:
            p.prove_file(d + "/m.wtns")
            stats["proofs"] += 1
        except k16.K16Error:
            stats["prove_err"] += 1
        p.close()
    print(json.dumps(stats))


def facade(cases, seed):
    """The same mutations through the C++ facade in a process of its own (tests/cpp/fullprover_harness.cpp): it must exit
    by itself -- state / error codes of fullprover.hpp -- never by a signal."""
    import zkey_builder as zb
    rs = np.random.RandomState(seed)
    d = tempfile.mkdtemp()
    toy_z = open(os.path.join(ROOT, "tests", "golden", "toy", "toy_1.zkey"), "rb").read()
    toy_w = open(os.path.join(ROOT, "tests", "golden", "toy", "toy.wtns"), "rb").read()
    exe = os.path.join(ROOT, "keyless-zk-proofs_amd", "fullprover_harness")
    outcomes, signals, t0 = {}, [], time.time()
    for c in range(cases):
        which = rs.randint(3)
        open(d + "/m.zkey", "wb").write(mutate(rs, toy_z, "z") if which != 1 else toy_z)
        open(d + "/m.wtns", "wb").write(mutate(rs, toy_w, "w") if which != 0 else toy_w)
        r = subprocess.run([exe, d + "/m.zkey", d + "/m.wtns", "1"], capture_output=True, text=True, timeout=300)
        if r.returncode < 0:
            signals.append({"case": c, "signal": -r.returncode})
        first = [l for l in r.stdout.splitlines() if l.startswith(("state=", "type="))]
        key = " ".join(l.split(" ms=")[0] for l in first)
        outcomes[key] = outcomes.get(key, 0) + 1
    print(json.dumps({"fuzz": "mutated toy .zkey / .wtns through the C++ FullProver facade (one process per case)", "cases": cases,
                      "seed": seed, "outcomes": outcomes, "killed_by_signal": signals, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if signals else 0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--facade":
        facade(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    batch, tot, crashes, t0 = 100, {"create_err": 0, "prove_err": 0, "proofs": 0}, [], time.time()
    for b in range(0, cases, batch):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(min(batch, cases - b)), str(seed * 1000 + b)],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, K16_NO_WARMUP="0"))
        if r.returncode != 0:
            crashes.append({"batch_seed": seed * 1000 + b, "rc": r.returncode, "stderr": r.stderr[-400:]})
            continue
        st = json.loads(r.stdout.strip().splitlines()[-1])
        for k in tot:
            tot[k] += st[k]
    print(json.dumps({"fuzz": "mutated .zkey / .wtns through create + prove_file", "cases": cases, "seed": seed, **tot,
                      "crashed_batches": crashes, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if crashes else 0)
