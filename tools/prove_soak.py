"""Soak of the full-size prover: P provers sharing the GPU prove the same four Keyless-shape witnesses over and over with fixed
blinding scalars; every proof must be the same bytes as the first proof of its witness (which the -m gpu suite compares with the
CPU oracle at this size).  Catches what a parity test of two proofs cannot: rare orderings between lanes, workspaces, graphs.
    python tools/prove_soak.py [proofs per prover] [provers]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
sys.path.insert(0, ROOT)
import k16  # noqa: E402
import bench  # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 500
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx0 = k16.Context(0)
n_vars, N, n_coefs = bench.KEYLESS["n_vars"], bench.KEYLESS["domain"], bench.KEYLESS["n_coefs"]
zpath = "/tmp/k16_soak_%d.zkey" % os.getpid()
with open(zpath, "wb") as f:
    f.write(bench.synth_zkey_bytes(ctx0, k16, n_vars, 1, N, n_coefs))
r, s = bench._le32(12345678901234567890 % bench.R_MOD), bench._le32(98765432109876543210 % bench.R_MOD)
wits = [bench.synth_witness(n_vars, 100 + i) for i in range(4)]
# round 6: the other provers share the first one's resident key (k16_prover_create_shared), as FullProver's K16_DEVICES=0,0 does
p0 = k16.Prover(ctx0, zpath)
share = None if os.environ.get("K16_SOAK_NO_SHARE") else p0
provers = [p0] + [k16.Prover(k16.Context(0), zpath, share_key_of=share) for _ in range(P - 1)]
ref = [provers[0].prove_mem(w, r, s) for w in wits]
bad = []


def worker(i):
    for k in range(per):
        j = (i + k) % len(wits)
        if provers[i].prove_mem(wits[j], r, s) != ref[j]:
            bad.append((i, k, j))


t0 = time.time()
th = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.time() - t0
print(json.dumps({"soak": "Keyless-shape proofs, %d provers sharing the GPU, fixed blinding" % P, "proofs": per * P,
                  "mismatches": bad[:10], "n_mismatches": len(bad), "proofs_per_s": round(per * P / dt, 1), "seconds": round(dt, 1)}))
os.unlink(zpath)
sys.exit(1 if bad else 0)
