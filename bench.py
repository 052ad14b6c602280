#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Groth16 hot path.

    python bench.py --gpus N --steps K --warmup W [--mode weak|strong] [--total-log2n 26] [--proofs P]

Headline (`value`, BASELINE.json configs[1]): BN254 G1 Pippenger MSM, 2^20 points per GPU, uniform random scalars in
[0, r), bases (i+1)*G -- all resident in HBM before the timed region.  One "step" = one complete MSM (digits -> bucket
sort -> bucket accumulation -> weighted bucket reduction -> Horner combine -> XYZZ result on the host).  Steps are
software-pipelined four deep over the context's MSM lanes (stream + workspace each); the timed region still contains
exactly K complete MSMs, each fully reduced to one point, and the last result is checked (untimed) against the closed
form  sum_i s_i (i+1) * G.

Multi-GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before anything touches the GPU and
exits with its return code (the driver's own torchrun launch is used as it is).  One rank per GPU over RCCL:
  --mode weak   (default) every rank owns an independent 2^20-point shard of an N*2^20-point MSM; each step ends with the
                path's one exchange, an all_gather of the 128-byte per-shard partial results + EC-add fold
  --mode strong BASELINE config 5: ONE MSM of 2^total-log2n points (default 2^26), rank k takes
                sharding.shard_range(2^26, N, k); same exchange; `scaling` = "strong"

The same JSON line also carries
  proof         BASELINE config 3 / 4: full Groth16 proofs of a synthetic Keyless-shape key (nVars 1,343,588, N = 2^21,
                8.3 M coefficients) on every rank (one prover per GPU, replicas): proofs/s over all ranks, p50 / p99
                latency through k16_prover_prove_mem, and (world size 1) through the file-based C++ FullProver facade
  config4_wave  BASELINE config 4 (round 6): a wave of 64 DISTINCT witnesses of a VALID synthetic Keyless-shape key, proof j on
                rank j mod N, then ONE k16_verify_batch of all 64 on rank 0
  strong_2p26   BASELINE config 5 (round 6): ONE 2^26-point G1 MSM -- eight shards through k16_msm_sharded_* at N = 1, one shard per
                rank + libk16.so's k16_rank_comm_* (ONE ncclAllGather) at N > 1 -- closed form checked
                (both after the timed region; K16_BENCH_NO_CONFIG_LEGS=1 skips them)
  roofline      for the dominant kernel (bucket accumulation): algorithmic bytes / measured launch time
  cpu_baseline  the CPU oracle (oracle/, a port of the reference algorithm) timed on this host on the same MSM workload
                and on ONE full proof of the same key, which is also the check of the GPU proof (rank 0, N = 1 only)
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))

LOG2N = 20
SHARE_GPU = False
XDEV = "cuda"
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ALGO_BYTES_PER_POINT = 64 + 32  # G1 MSM = n x (64 B affine point + 32 B scalar)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
MODMUL_PER_PAIR = 10            # SURVEY 8(d) secondary figure: one mixed add (8M + 2S) per (point, window) pair
MODMUL_PEAK_G = 174.3           # measured on MI355X: radix-2^29 Montgomery multiply (fmul29 chains of tools/ubench.hip, 8 blocks
                                # per CU), all CUs: profiles/r02/ubench_instruction_rates.log
# The same ceiling from the HARDWARE's rate instead of this repository's own multiplication loop: v_mad_u64_u32 issues at
# 31.9 T lane-operations/s chip-wide (profiles/r05/ubench2_instruction_costs.log, 8 waves per SIMD) and a 254-bit Montgomery
# product on 29-bit limbs needs at least 162 of them (81 product + 81 reduction terms) -- every other instruction of fmul29
# (masks, shifts, the nine m_k) counts against this figure.
MAD_U64_LANE_OPS_T = 31.9
MODMUL_PEAK_HW_G = MAD_U64_LANE_OPS_T * 1e3 / 162.0     # 196.9 G modmul/s
KEYLESS = dict(n_vars=1343588, n_public=1, domain=1 << 21, n_coefs=8300000)   # circuit/README.md:77-83, SURVEY 8(d)


def uniform_scalars(n, seed):
    """n x 32 B little-endian, uniform in [0, r) by rejection on 254-bit draws."""
    rs = np.random.RandomState(seed)
    out = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)
    out[:, 31] &= 0x3F
    r_be = np.frombuffer(R_MOD.to_bytes(32, "big"), dtype=np.uint8)
    while True:
        be = out[:, ::-1]
        diff = be != r_be
        first = diff.argmax(axis=1)
        idx = np.arange(n)
        ge = np.where(diff.any(axis=1), be[idx, first] > r_be[first], True)
        bad = np.nonzero(ge)[0]
        if bad.size == 0:
            return np.ascontiguousarray(out)
        rep = rs.randint(0, 256, size=(bad.size, 32), dtype=np.uint8)
        rep[:, 31] &= 0x3F
        out[bad] = rep


def fast_scalars(n, seed):
    """n x 32 B, uniform in [0, 2^253) (always < r): the generator for the 2^26-point strong-scaling leg, where the
    rejection sampler above would take minutes on the host."""
    g = np.random.default_rng(seed)
    out = g.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << 61) - 1)
    return out.view(np.uint8).reshape(n, 32)


def weighted_sum_mod_r(scalars, start):
    """sum_i s_i * (start + i + 1) mod r, exactly: the scalar k with MSM(s, bases (start+i+1)G) = k*G.
    16-bit limbs of s times 14-bit halves of the weight, summed in u64 per 2^20-row chunk (no overflow: < 2^50)."""
    n = scalars.shape[0]
    total = 0
    for lo in range(0, n, 1 << 20):
        hi = min(n, lo + (1 << 20))
        limbs = scalars[lo:hi].view("<u2").reshape(hi - lo, 16).astype(np.uint64)
        w = np.arange(start + lo + 1, start + hi + 1, dtype=np.uint64)
        for shift, part in ((0, w & np.uint64(0x3FFF)), (14, (w >> np.uint64(14)) & np.uint64(0x3FFF)), (28, w >> np.uint64(28))):
            if not part.any():
                continue
            cols = part @ limbs           # 16 exact column sums
            total += sum(int(cols[k]) << (16 * k) for k in range(16)) << shift
    return total % R_MOD


def scalar_times_g(ctx, k16, k):
    """k*G through the library's own n = 1 path (Curve::mulByScalar's replacement) -- the closed-form side of the check."""
    d_g = ctx.synth_points(k16.G1, 0, 1)
    d_k = ctx.to_device(np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
    _, aff = ctx.msm_device(k16.G1, d_g, d_k, 1)
    d_g.free()
    d_k.free()
    return aff


# ---------------------------------------------------------------------------------------------------- CPU baseline
KERNEL_SOURCES = ("csrc/msm_kernels.inc", "csrc/bn254_fq9.h", "csrc/bn254_field.h", "csrc/bn254_curve.h", "csrc/msm_g1.hip")


def kernel_sources_sha16():
    """Digest of the files that make up k_accumulate<Eng9> (what profiles/pmc_traffic.json's counters belong to)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "keyless-zk-proofs_amd", rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def usable_cpus():
    """CPUs this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one (os.cpu_count() reports
    the machine's cores even inside a container that is allowed a few of them -- 64 OpenMP threads on 16 allowed cores is
    what made round 2's CPU baseline scale so badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(n_sample, scalars):
    """Times the CPU oracle (port of the reference's ParallelMultiexp) on the first n_sample points."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol  # the checker, used here only as the reported CPU baseline

    # the port parallelises over (window, point slice) pairs like the reference's TBB loop; at 2^20 points it has 16 windows
    # x 4 slices = 64 tasks (more slices cost more in the pack step than they spread), so more threads would only idle
    avail = usable_cpus()
    bases = ol.gen_points(0, 0, n_sample)
    sc = np.ascontiguousarray(scalars[:n_sample])
    ol.msm(0, bases[:4096], sc[:4096], nthreads=min(avail, 16))  # warm-up
    # at 2^20 points the port has 16 windows x 4 point slices = 64 tasks (more slices cost more in the pack step than they
    # spread): 16, 32 and 64 threads are tried (as far as the host allows) and the best is reported with ITS thread count
    best = None
    for threads in sorted({min(avail, t) for t in (16, 32, 64)}):
        t0 = time.time()
        ol.msm(0, bases, sc, nthreads=threads)
        dt = time.time() - t0
        if best is None or n_sample / dt > best[0]:
            best = (n_sample / dt, threads, dt)
    value, threads, dt = best
    return {
        "value": value,
        "unit": "points/s",
        "cores": threads,
        "usable_cpus": avail,
        "points_per_s_per_core": value / threads,
        "reference_single_thread_anchor": "36-37 k points/s/core (SURVEY section 6: the reference's generic backend, one thread)",
        "kind": "port",
        "sample": "first 2^%d points of the same workload, %.1f s wall at the best of 16 / 32 / 64 threads (of %d usable CPUs), "
                  "oracle/bn254_ref.c (gcc -O2, OpenMP over windows x point slices, per-task bucket arrays + pack as "
                  "multiexp.cpp:46-130)" % (int(np.log2(n_sample)), dt, avail),
    }


# ---------------------------------------------------------------------------------------------------- proof leg
def _le32(x):
    return int(x).to_bytes(32, "little")


def _section(t, payload):
    return struct.pack("<IQ", t, len(payload)) + payload


def synth_zkey_bytes(ctx, k16, n_vars, n_public, N, n_coefs, seed=1):
    """Synthetic zkey of the Keyless SHAPE in the iden3 container the prover parses: points are
    multiples of the generators made on the device, B1/B2 half (0,0), coefficients small values * R^2 in snarkjs order."""
    rs = np.random.RandomState(seed)

    def pts(group, start, n, zero_frac=0.0):
        d = ctx.synth_points(group, start, n)
        a = d.download(np.uint8, (n, k16.AFF_BYTES[group])).copy()
        d.free()
        if zero_frac > 0:
            a[rs.rand(n) < zero_frac] = 0
        return a.tobytes()

    g1 = pts(k16.G1, 100, 3)
    g2 = pts(k16.G2, 50, 3)
    hdr = struct.pack("<I", 32) + _le32(Q_MOD) + struct.pack("<I", 32) + _le32(R_MOD) + struct.pack("<III", n_vars, n_public, N)
    hdr += g1[0:64] + g1[64:128] + g2[0:128] + g2[128:256] + g1[128:192] + g2[256:384]
    coef = np.zeros(n_coefs, dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "V32")])
    coef["m"] = rs.randint(0, 2, size=n_coefs)
    coef["c"] = np.sort(rs.randint(0, N, size=n_coefs))
    coef["s"] = rs.randint(0, n_vars, size=n_coefs)
    r2 = pow(1 << 256, 2, R_MOD)
    table = np.frombuffer(b"".join(_le32(v * r2 % R_MOD) for v in range(1, 257)), dtype="V32")
    coef["v"] = table[rs.randint(0, 256, size=n_coefs)]
    secs = [_section(1, struct.pack("<I", 1)), _section(2, hdr),
            _section(4, struct.pack("<I", n_coefs) + coef.tobytes()),
            _section(5, pts(k16.G1, 1000, n_vars)),
            _section(6, pts(k16.G1, 3000000, n_vars, 0.5)),
            _section(7, pts(k16.G2, 5000, n_vars, 0.5)),
            _section(8, pts(k16.G1, 6000000, n_vars - n_public - 1)),
            _section(9, pts(k16.G1, 9000000, N))]
    return b"zkey" + struct.pack("<II", 1, len(secs)) + b"".join(secs)


def synth_witness(n_vars, seed):
    """90 % bits, 8 % bytes, 2 % full-width values; w[0] = 1."""
    rs = np.random.RandomState(seed)
    w = np.zeros((n_vars, 32), dtype=np.uint8)
    u = rs.rand(n_vars)
    bits = u < 0.90
    w[bits, 0] = rs.randint(0, 2, size=int(bits.sum()))
    byts = (u >= 0.90) & (u < 0.98)
    w[byts, 0] = rs.randint(0, 256, size=int(byts.sum()))
    full = u >= 0.98
    f = rs.randint(0, 256, size=(int(full.sum()), 32), dtype=np.uint8)
    f[:, 31] &= 0x1F
    w[full] = f
    w[0] = 0
    w[0, 0] = 1
    return w


def write_wtns(path, w):
    sec1 = struct.pack("<I", 32) + _le32(R_MOD) + struct.pack("<I", w.shape[0])
    with open(path, "wb") as f:
        f.write(b"wtns" + struct.pack("<II", 2, 2) + _section(1, sec1) + _section(2, w.tobytes()))


def facade_leg(zpath, wpath, proofs, mem=False):
    """mem=True: the same FullProver object through k16_fullprover_prove_mem (the witness handed over in memory).
    The file-based drop-in boundary: FullProver(zkey).prove(wtns_path) in a C++ process (tests/cpp/fullprover_harness.cpp,
    what the Rust crate does through bindgen), timed by the harness around its prove() loop."""
    pkg = os.path.join(ROOT, "keyless-zk-proofs_amd")
    exe = os.path.join(pkg, "fullprover_harness")
    if not os.path.exists(exe):      # build() makes it; fall back to compiling it here
        exe = "/tmp/k16_fullprover_harness_%d" % os.getpid()
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "fullprover_harness.cpp"), "-L", pkg, "-lk16",
                               "-Wl,-rpath," + pkg, "-pthread", "-o", exe])
    env = dict(os.environ, K16_HARNESS_MEM="1") if mem else None
    out = subprocess.run([exe, zpath, wpath, str(proofs + 1)], capture_output=True, text=True, timeout=600, env=env)
    lines = out.stdout.splitlines()
    ms = [int(l.split("ms=")[1]) for l in lines if l.startswith("type=0 error=0")]
    tot = [l for l in lines if l.startswith("elapsed_ms=")]
    if out.returncode != 0 or len(ms) != proofs + 1 or not tot:
        return {"error": "harness rc=%d: %s" % (out.returncode, (out.stderr or out.stdout)[-300:])}
    # the first prove of a fresh process allocates the MSM workspaces: reported, not averaged in
    elapsed = float(tot[0].split()[0].split("=")[1])
    steady = (elapsed - ms[0]) if elapsed > ms[0] else elapsed
    if mem:
        return {"proofs_per_s": proofs / (steady * 1e-3), "proofs": proofs,
                "note": "k16_fullprover_prove_mem on the same FullProver object: the witness values in memory, no file"}
    return {"proofs_per_s": proofs / (steady * 1e-3), "prover_time_ms_p50": float(np.median(ms[1:])),
            "first_prove_ms": ms[0], "proofs": proofs,
            "note": "FullProver(zkey).prove(path): mmap + parse of the 43 MB .wtns file inside every call; prover_time is "
                    "the reference's own metric (RS/fullprover.cpp:226-244, whole milliseconds)"}


def facade_pool_leg(zpath, wpath, proofs_per_thread):
    """Throughput mode THROUGH THE DROP-IN BOUNDARY: one FullProver whose pool holds two provers on this GPU (K16_DEVICES=0,0 -- one
    resident key, the second prover one placeholder stream behind the first, yielding waits: what fullprover.cpp sets up by itself),
    two caller threads, the witness handed over in memory (k16_fullprover_prove_mem) -- the C++ process a Rust service with two
    workers per GPU is.  prover_handler.rs:244-345 with the mutex of prover_state.rs:21 replaced by the pool."""
    pkg = os.path.join(ROOT, "keyless-zk-proofs_amd")
    exe = os.path.join(pkg, "fullprover_harness")
    if not os.path.exists(exe):
        return {"error": "fullprover_harness not built"}
    env = dict(os.environ, K16_DEVICES="0,0", K16_HARNESS_MEM="1")
    out = subprocess.run([exe, zpath, wpath, str(proofs_per_thread), "2"], capture_output=True, text=True, timeout=900, env=env)
    lines = out.stdout.splitlines()
    ok = sum(1 for l in lines if l.startswith("type=0 error=0"))
    tot = [l for l in lines if l.startswith("elapsed_ms=")]
    if out.returncode != 0 or ok != 2 * proofs_per_thread or not tot:
        return {"error": "harness rc=%d, %d proofs ok: %s" % (out.returncode, ok, (out.stderr or out.stdout)[-300:])}
    elapsed = float(tot[0].split()[0].split("=")[1])
    return {"proofs_per_s": ok / (elapsed * 1e-3), "proofs": ok, "provers": 2, "caller_threads": 2,
            "note": "FullProver with K16_DEVICES=0,0 (one resident key, stream offset, yielding waits), k16_fullprover_prove_mem from two "
                    "threads; includes each prover's first proof"}


def verify_leg(ctx, k16, with_oracle):
    """The batched GPU verifier (k16_verify_batch) on toy-circuit proofs made by the GPU prover: the only
    key with a verification key offline; the cost of a Groth16 check does not depend on the circuit (3 pairings + one
    scalar multiplication per public input; Keyless has one public input, like the toy circuit)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import groth16_io as gio
    toy = os.path.join(ROOT, "tests", "golden", "toy")
    vk = gio.vk_from_json(os.path.join(toy, "toy_vk.json"))
    V = k16.VerifyingKey(ctx, vk)
    p = k16.Prover(ctx, os.path.join(toy, "toy_1.zkey"))
    proofs = [gio.proof_from_json(p.prove_file(os.path.join(toy, "toy.wtns"))) for _ in range(8)]
    p.close()
    out = {"entry": "k16_verify_batch", "key": "toy circuit (tests/golden/toy/toy_vk.json), 1 public input", "batches": []}
    for n in (1, 64, 4096):
        pr = [proofs[i % 8] for i in range(n)]
        inp = [[2 if i % 5 else 3] for i in range(n)]         # every fifth proof gets the wrong public input
        V.verify_batch(pr, inp)
        t0 = time.perf_counter()
        ok = V.verify_batch(pr, inp)
        ms = (time.perf_counter() - t0) * 1e3
        if ok != [bool(i % 5) for i in range(n)]:
            raise SystemExit("bench.py: GPU verifier accepted / rejected the wrong proofs")
        out["batches"].append({"n": n, "ms": ms, "proofs_per_s": n / ms * 1e3})
    out["checked"] = True
    if with_oracle:
        import oracle_lib as ol
        t0 = time.perf_counter()
        for pr in proofs[:4]:
            assert ol.groth16_verify(vk, pr, [2])
        out["cpu_oracle_ms_per_proof"] = (time.perf_counter() - t0) / 4 * 1e3
    V.close()
    return out


def ctx_device(ctx):
    return getattr(ctx, "device", 0)


def proof_leg(ctx, k16, torch, dist, rank, world, proofs, check_with_oracle, scale=1.0):
    n_vars = max(int(KEYLESS["n_vars"] * scale), 8)
    N = 1 << max(int(np.ceil(np.log2(max(1376867 * scale, 4)))), 2)
    n_coefs = int(KEYLESS["n_coefs"] * scale)
    t0 = time.time()
    # one key file per job: rank 0 writes it, every rank of the node loads it (0.93 GB at Keyless shape)
    tag = os.environ.get("MASTER_PORT", str(os.getpid())) if dist is not None else str(os.getpid())
    zpath = "/tmp/k16_bench_%s.zkey" % tag
    if rank == 0:
        zk = synth_zkey_bytes(ctx, k16, n_vars, 1, N, n_coefs)
        with open(zpath + ".part", "wb") as f:
            f.write(zk)
        os.replace(zpath + ".part", zpath)
        del zk
    if dist is not None:
        dist.barrier()
    t_key = time.time() - t0
    t0 = time.time()
    prover = k16.Prover(ctx, zpath)
    t_create = time.time() - t0
    r, s = _le32(12345678901234567890 % R_MOD), _le32(98765432109876543210 % R_MOD)
    wits = [synth_witness(n_vars, 100 + 16 * rank + i) for i in range(4)]
    prover.prove_mem(wits[0], r, s)      # warm-up: workspace allocation
    prover.prove_mem(wits[1], r, s)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = []
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t_all = time.perf_counter()
    for i in range(proofs):
        t1 = time.perf_counter()
        prover.prove_mem(wits[i % len(wits)])          # production path: blinding drawn from the OS CSPRNG
        lat.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_all
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    # host CPU time one proof costs this PROCESS, all threads (witness scan on the library's pool, launches, the MSMs' host
    # combines, blinding, JSON): what a node's host has to supply per proof and per GPU beside the GPU time
    host_cpu_ms = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) * 1e3 / max(proofs, 1)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=XDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        allv = [None] * world
        dist.all_gather_object(allv, lat)
        lat = [x for l in allv for x in l]
    # throughput mode (rank 0 reports, every rank runs it so that the GPUs stay symmetric): two provers -- own context,
    # streams and resident key each -- share this GPU and prove concurrently from two threads, the deployment of
    # INTEGRATION.md section 4 (K16_DEVICES=0,0 behind one FullProver).  One proof's upload / chain / H MSM then runs
    # under another's; latency per proof rises, proofs per second too.
    thr = None
    # measured, round 4 (profiles/r04/throughput_provers_sweep.log): 2 -> 180-184, 3 -> 175-178, 4 -> 170-172, 6 / 8 -> 170 proofs/s,
    # with p50 10.5 / 17 / 23.5 ms: two provers fill the chip (2.16-2.18 GHz at 1140 W, profiles/r04/clocks_under_proof_load.log)
    n_conc = int(os.environ.get("K16_BENCH_PROVERS", "2"))
    if n_conc > 1:
        import threading
        # (the further provers of this GPU sit i placeholder streams behind the first, as in FullProver's pool: k16_ctx_create_ex)
        off = int(os.environ.get("K16_BENCH_OTHER_OFFSET", "1"))
        others = [k16.Context(ctx_device(ctx), stream_offset=((i + 1) * off) % 4) for i in range(n_conc - 1)]
        for c in [ctx] + others:
            c.set_option(k16.OPT_SHARED_GPU, 1)      # what FullProver does for K16_DEVICES=0,0 (include/k16.h)
        provers = [prover] + [k16.Prover(c, zpath) for c in others]
        for pv in provers[1:]:
            pv.prove_mem(wits[0], r, s)
            pv.prove_mem(wits[1], r, s)
        lats = [[] for _ in provers]
        # at least 60 proofs per prover (a third of a second each): with the latency leg's 20 the leg is over before the
        # clocks and the two provers' interleaving have settled (171-179 against 181-190 proofs/s for the same build)
        thr_proofs = max(proofs, 60)

        def worker(i):
            for k in range(thr_proofs):
                t1 = time.perf_counter()
                provers[i].prove_mem(wits[(i + k) % len(wits)])
                lats[i].append((time.perf_counter() - t1) * 1e3)

        def timed_wave():
            for l in lats:
                del l[:]
            if dist is not None:
                dist.barrier()
            r0 = resource.getrusage(resource.RUSAGE_SELF)
            t_all = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i,)) for i in range(n_conc)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            el = time.perf_counter() - t_all
            r1 = resource.getrusage(resource.RUSAGE_SELF)
            cpu_ms = ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) * 1e3 / (n_conc * thr_proofs)
            allv = [x for l in lats for x in l]
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device=XDEV)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return {"provers_per_gpu": n_conc, "proofs_per_s": world * n_conc * thr_proofs / el, "p50_ms": float(np.median(allv)),
                    "p99_ms": float(np.percentile(allv, 99)), "proofs": world * n_conc * thr_proofs, "host_cpu_ms_per_proof": cpu_ms}

        thr = timed_wave()
        # the same wave with K16_OPT_YIELDING_WAITS (what FullProver sets for a pool): the callers' waits poll + sleep instead of
        # spinning inside the runtime -- same rate, a fraction of the host CPU time per proof (a node runs 16 such callers)
        for c in [ctx] + others:
            c.set_option(k16.OPT_YIELDING_WAITS, 1)
        thr["yielding_waits"] = timed_wave()
        for c in [ctx] + others:
            c.set_option(k16.OPT_YIELDING_WAITS, 0)
        for pv in provers[1:]:
            pv.close()
        for c in others:
            c.close()
        ctx.set_option(k16.OPT_SHARED_GPU, 0)
    out = None
    if rank == 0:
        p50 = float(np.median(lat))
        # SURVEY 8(d) for the full proof.  Multiplications (254-bit modular, the unit that bounds the path): H MSM n x 13
        # digit positions x 10; six transforms of N/2 log2 N butterflies + ~5 N pointwise; witness MSMs by the expected
        # non-zero digits of the 90 / 8 / 2 % witness mix at c = 13 (0.93 per wire; B1 / B2 have half (0,0) rows; a G2
        # mixed add is 10 Fq2 = 30 Fq multiplications); bucket reductions 2 full adds (14) per bucket; SpMV one per coefficient.
        logN = int(np.log2(N))
        mm = {"h_msm": N * 13 * 10, "ntt_and_pointwise": 6 * (N // 2) * logN + 5 * N,
              "witness_msm_g1": int(n_vars * 0.93 * (1 + 0.5 + 1) * 10), "witness_msm_g2": int(n_vars * 0.93 * 0.5 * 30),
              "bucket_reductions": (1 << 19) * 2 * 14 + 21 * 4096 * 2 * 14 * (3 + 3), "spmv": n_coefs}
        mm_total = sum(mm.values())
        hbm_bytes = (n_vars * (64 + 64 + 128 + 64) + N * 64) + n_vars * 32 + N * 32 + 6 * 2 * 32 * N + 44 * n_coefs + 5 * 3 * 32 * N
        roof = {"modmul_per_proof": mm_total, "modmul_breakdown": mm,
                "alu_frac": mm_total / (p50 * 1e-3) / 1e9 / MODMUL_PEAK_G, "alu_peak_g_modmul_s": MODMUL_PEAK_G,
                "alu_frac_hw": mm_total / (p50 * 1e-3) / 1e9 / MODMUL_PEAK_HW_G, "alu_peak_hw_g_modmul_s": MODMUL_PEAK_HW_G,
                "hbm_bytes_per_proof": hbm_bytes, "hbm_frac": hbm_bytes / (p50 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "basis": "algorithmic work of one proof (SURVEY 8(d)) over the p50 latency of one proof at a time; the path is "
                         "integer-multiply-issue bound (alu_frac is the figure to move), hbm_frac is reported because "
                         "BASELINE.json's bound is HBM"}
        out = {"proofs_per_s": world * proofs / elapsed, "p50_ms": p50, "roofline": roof,
               "p99_ms": float(np.percentile(lat, 99)), "proofs": world * proofs, "entry": "k16_prover_prove_mem",
               "n_vars": n_vars, "domain": N, "n_coefs": n_coefs, "n_public": 1, "host_cpu_ms_per_proof": host_cpu_ms,
               "key": "synthetic, Keyless shape (the real zkey is not available offline); one resident copy per GPU",
               "parallelism": "replicas: one prover per GPU, no collective" if world > 1 else "single GPU",
               "setup_s": {"synthesize_key": t_key, "prover_create": t_create}, "checked": None,
               "throughput_mode": thr}
        wpath = "/tmp/k16_bench_%d.wtns" % os.getpid()
        write_wtns(wpath, wits[0])
        got = prover.prove_mem(wits[0], r, s)
        # secondary figure: the compact hand-off (k16_prover_prove_compact, include/k16.h) -- the witness written in the
        # device's upload form (one byte per wire + the list of the wide values) into the prover's pinned buffers OUTSIDE
        # the timed call, as a witness calculator that owns its output would: the proof without the host scan of the 43 MB
        # array.  The headline p50 above stays the k16_prover_prove_mem one (the reference's hand-off is a full witness).
        try:
            def fill(w):
                w2 = np.ascontiguousarray(w, dtype=np.uint8).reshape(-1, 32)
                narrow, idx, val = prover.compact_buffers()
                wide = np.flatnonzero(w2[:, 1:].any(axis=1))
                narrow[:] = w2[:, 0]
                narrow[wide] = 0
                idx[:len(wide)] = wide
                val[:len(wide)] = w2[wide]
                return len(wide)
            same = prover.prove_compact(fill(wits[0]), r, s) == got
            clat = []
            nw = fill(wits[1])                 # (filled once: refilling from Python between proofs leaves the GPU idle for
            prover.prove_compact(nw)           #  ~0.1 s each time and measures its clocks ramping up, not the hand-off)
            for i in range(proofs):
                t1 = time.perf_counter()
                prover.prove_compact(nw)
                clat.append((time.perf_counter() - t1) * 1e3)
            out["compact_hand_off"] = {"entry": "k16_prover_prove_compact", "p50_ms": float(np.median(clat)),
                                       "p99_ms": float(np.percentile(clat, 99)), "proofs": proofs,
                                       "same_proof_as_prove_mem": bool(same),
                                       "note": "witness already in the prover's pinned upload buffers in compact form when the call starts"}
        except Exception as e:
            out["compact_hand_off"] = {"error": repr(e)}
        try:
            # one proof at a time with K16_OPT_YIELDING_WAITS: what the waits' sleeps cost a proof's latency, and save its host
            ctx.set_option(k16.OPT_YIELDING_WAITS, 1)
            ylat = []
            r0 = resource.getrusage(resource.RUSAGE_SELF)
            for i in range(proofs):
                t1 = time.perf_counter()
                prover.prove_mem(wits[i % len(wits)])
                ylat.append((time.perf_counter() - t1) * 1e3)
            r1 = resource.getrusage(resource.RUSAGE_SELF)
            out["yielding_waits"] = {"p50_ms": float(np.median(ylat)), "p99_ms": float(np.percentile(ylat, 99)),
                                     "host_cpu_ms_per_proof": ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) * 1e3 / max(proofs, 1)}
        except Exception as e:
            out["yielding_waits"] = {"error": repr(e)}
        finally:
            ctx.set_option(k16.OPT_YIELDING_WAITS, 0)
        if world == 1:
            try:
                out["facade"] = facade_leg(zpath, wpath, max(4, proofs // 2))
                out["facade_mem"] = facade_leg(zpath, wpath, max(4, proofs // 2), mem=True)
                out["facade_pool"] = facade_pool_leg(zpath, wpath, max(60, proofs))
            except Exception as e:
                out["facade"] = {"error": repr(e)}
        if check_with_oracle:
            # ONE full proof by the CPU oracle on the same key, witness and injected (r, s): the CPU prover's time next to
            # the GPU's, and the byte-for-byte check of the GPU proof
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as ol
            threads = min(usable_cpus(), 96)
            t0 = time.time()
            want = ol.prove_files(zpath, wpath, r, s, nthreads=threads)
            cpu_s = time.time() - t0
            out["checked"] = bool(got == want)
            out["cpu_oracle"] = {"seconds_per_proof": cpu_s, "proofs_per_s": 1.0 / cpu_s, "cores": threads, "kind": "port",
                                 "sample": "one full proof of the same key and witness (oracle/bn254_ref.c groth16_prove)"}
            if got != want:
                raise SystemExit("bench.py: GPU proof differs from the CPU oracle's")
        os.unlink(wpath)
        try:
            # the service's sequence: prove, then verify THAT proof before releasing it (prover_handler.rs:329-336).  The
            # verification's cost does not depend on the circuit or on the proof being valid (3 Miller loops + one final
            # exponentiation; one public input like Keyless): a toy-key proof stands in for the check of the synthetic key's.
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import groth16_io as gio
            toy = os.path.join(ROOT, "tests", "golden", "toy")
            V = k16.VerifyingKey(ctx, gio.vk_from_json(os.path.join(toy, "toy_vk.json")))
            tp = k16.Prover(ctx, os.path.join(toy, "toy_1.zkey"))
            toy_proof = gio.proof_from_json(tp.prove_file(os.path.join(toy, "toy.wtns")))
            tp.close()
            assert V.verify_batch([toy_proof], [[2]]) == [True]
            both = []
            for i in range(min(proofs, 12)):
                t1 = time.perf_counter()
                prover.prove_mem(wits[i % len(wits)])
                ok = V.verify_batch([toy_proof], [[2]])
                both.append((time.perf_counter() - t1) * 1e3)
                assert ok == [True]
            out["p50_with_verify_ms"] = float(np.median(both))
            V.close()
        except Exception as e:
            out["p50_with_verify_ms"] = None
            out["p50_with_verify_error"] = repr(e)
        try:
            out["verify"] = verify_leg(ctx, k16, check_with_oracle)
        except SystemExit:
            raise
        except Exception as e:
            out["verify"] = {"error": repr(e)}
    prover.close()
    if dist is not None:
        dist.barrier()          # nobody is still opening the key
    if rank == 0:
        os.unlink(zpath)
    return out



# ---------------------------------------------------------------------------------------------------- BASELINE configs 4 and 5
def _tag(dist):
    return os.environ.get("MASTER_PORT", str(os.getpid())) if dist is not None else str(os.getpid())


def make_rank_exchange(dist, ctx, sharding):
    """The library's RCCL leg (k16_rank_comm_*) on every rank or on none; (exchange, note).  Real RCCL refuses two ranks on one
    device, so the one-GPU test rig (K16_BENCH_SHARE_GPU) gets it only over the test double named by K16_RCCL_LIB."""
    import torch
    if dist is None or os.environ.get("K16_BENCH_EXCHANGE", "c") != "c" or (SHARE_GPU and not os.environ.get("K16_RCCL_LIB")):
        return None, None
    ex, note = None, None
    try:
        ex = sharding.RankExchange(dist, ctx)
    except Exception as e:
        note = "C exchange unavailable (%r): torch.distributed all_gather instead" % (e,)
        print("bench.py: WARNING: " + note, file=sys.stderr, flush=True)
    ok = torch.tensor([1 if ex is not None else 0], device=XDEV)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # all ranks or none
    if int(ok.item()) == 0 and ex is not None:
        ex.close()
        ex = None
    return ex, note


def all_ranks_ok(torch, dist, ok):
    """Agreement before a leg's first collective: a rank whose LOCAL set-up failed (allocation, key load ...) must not leave the
    others waiting inside a collective it will never join -- every rank learns of it here and the leg is skipped everywhere."""
    if dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], device=XDEV)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def strong_leg(k16, torch, dist, sharding, rank, world, dev, log2n):
    """BASELINE config 5: ONE G1 MSM of 2^log2n points (default 2^26: 4 GiB of points, 2 GiB of scalars) cut N ways.
    N = 1: k16_msm_sharded_* with EIGHT shards over the visible devices (eight contexts on device 0 on a one-GPU box), host-side
    fold.  N > 1: rank k owns sharding.shard_range(2^log2n, N, k) as a one-shard k16_msm_sharded object on its GPU (the same
    pieced upload pipeline), and the ranks' partials travel through k16_rank_comm_* -- ONE ncclAllGather issued by libk16.so --
    and are folded on every rank.  The scalars arrive in HOST memory inside every run (Curve::multiMulByScalar's contract,
    curve.hpp:209-215 / multiexp.cpp:183-245); `device_scalars` is the same MSM with the scalars already resident.
    Checked against the closed form sum_i s_i (i+1) * G over ALL ranks' rows."""
    total = 1 << log2n
    lo, hi = sharding.shard_range(total, world, rank)
    n = hi - lo
    if world == 1:
        ndev = max(1, k16.load().k16_device_count())
        devices = [r % ndev for r in range(8)]
    else:
        devices = [dev]
    sm, ex, note, setup_err = None, None, None, None
    try:                                                   # local set-up: no collective in here
        t0 = time.time()
        scalars = fast_scalars(n, seed=0x2626 + rank)
        t_scal = time.time() - t0
        sm = k16.ShardedMsm(devices, k16.G1, n)
        t0 = time.time()
        for r in range(sm.count()):
            slo, shi = sm.shard_range(r)
            d = sm.shard_ctx(r).synth_points(k16.G1, lo + slo, shi - slo)
            sm.set_bases_device(r, d)
            d.free()
        t_bases = time.time() - t0
    except Exception as e:
        setup_err = repr(e)
    if not all_ranks_ok(torch, dist, setup_err is None):
        if sm is not None:
            sm.close()
        return {"error": "set-up failed on a rank (%s): leg skipped on all ranks" % (setup_err or "another rank")} if rank == 0 else None
    try:
        ex, note = make_rank_exchange(dist, sm.shard_ctx(0), sharding)

        def fold(xyzz):
            if ex is not None:
                return ex.exchange_and_fold(k16.G1, xyzz)[0]
            if dist is not None:
                return sharding.exchange_and_fold(dist, k16.G1, xyzz, device=None if SHARE_GPU else "cuda")[0]
            return xyzz

        def timed(run, reps):
            best = None
            for _ in range(reps):
                if dist is not None:
                    dist.barrier()
                t1 = time.perf_counter()
                xyzz = run()
                t2 = time.perf_counter()
                res = fold(xyzz)
                t3 = time.perf_counter()
                if dist is not None:
                    dist.barrier()
                t4 = time.perf_counter()
                cur = {"total_ms": (t4 - t1) * 1e3, "shard_ms": (t2 - t1) * 1e3, "exchange_and_fold_ms": (t3 - t2) * 1e3,
                       "one_process_fold_ms": sm.last_ms()["fold_ms"]}
                if dist is not None:
                    t = torch.tensor([cur["total_ms"], cur["shard_ms"], cur["exchange_and_fold_ms"]], dtype=torch.float64, device=XDEV)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    cur["total_ms"], cur["shard_ms"], cur["exchange_and_fold_ms"] = (float(v) for v in t.tolist())
                if best is None or cur["total_ms"] < best["total_ms"]:
                    best = cur
            return res, best

        sm.run(scalars)                                   # warm-up: the lanes' workspaces
        res_host, host = timed(lambda: sm.run(scalars)[0], 3)
        # ... the same host array page-locked by the caller (k16_host_register): the uploads are then DMA copies
        pinned_t, res_pin = None, None
        try:
            sm.shard_ctx(0).host_register(scalars)
            try:
                res_pin, pinned_t = timed(lambda: sm.run(scalars)[0], 3)
            finally:
                sm.shard_ctx(0).host_unregister(scalars)
        except Exception as e:
            pinned_t = {"error": repr(e)}
        ds = []
        for r in range(sm.count()):
            slo, shi = sm.shard_range(r)
            ds.append(sm.shard_ctx(r).to_device(scalars[slo:shi]))
        res_dev, devt = timed(lambda: sm.run_device(ds)[0], 3)
        for d in ds:
            d.free()
        t0 = time.time()
        k_mine = weighted_sum_mod_r(scalars, lo)
        t_check = time.time() - t0
        ranks_seen = 1
        if dist is not None:
            ranks_seen = dist.get_world_size()
            ks = [None] * ranks_seen
            dist.all_gather_object(ks, k_mine)
            k_all = sum(ks) % R_MOD
        else:
            k_all = k_mine
        out = None
        if rank == 0:
            want = scalar_times_g(sm.shard_ctx(0), k16, k_all)
            ok = [bool(k16.points_sum(k16.G1, np.frombuffer(x, dtype=np.uint8).reshape(1, 128))[1] == want) for x in (res_host, res_dev)]
            out = {"workload": "ONE BN254 G1 MSM of 2^%d points, scalars uniform in [0, 2^253), bases (i+1)*G made on the devices" % log2n,
                   "entry": ("k16_msm_sharded_run: %d shards, one process, host-side EC-add fold" % sm.count()) if world == 1 else
                            ("k16_msm_sharded_run per rank (1 shard) + %s" %
                             ("k16_rank_comm_allgather_fold (ONE ncclAllGather issued by libk16.so)" if ex is not None else
                              "torch.distributed all_gather (gloo test rig)" if SHARE_GPU else "torch.distributed all_gather (RCCL)")),
                   "ranks_seen": ranks_seen, "shards": sm.count() * ranks_seen, "devices": devices if world == 1 else "one per rank",
                   "host_scalars": dict(host, points_per_s=total / (host["total_ms"] * 1e-3),
                                        note="scalars handed over in pageable host memory inside every run (PCIe-inclusive)"),
                   "host_scalars_page_locked": (dict(pinned_t, points_per_s=total / (pinned_t["total_ms"] * 1e-3),
                                                     same_result=bool(k16.points_sum(k16.G1, np.frombuffer(res_pin, dtype=np.uint8).reshape(1, 128))[1] == want),
                                                     note="the same array after k16_host_register (the caller's one-time cost, outside the run)")
                                                if pinned_t and "total_ms" in pinned_t else pinned_t),
                   "device_scalars": dict(devt, points_per_s=total / (devt["total_ms"] * 1e-3)),
                   "result_checked": all(ok), "exchange_note": note,
                   "setup_s": {"scalars": t_scal, "bases_on_device": t_bases, "closed_form_on_host": t_check}}
            if not all(ok):      # a SECONDARY leg: say so loudly, keep the line (the headline's own check still ends the run)
                out["error"] = "the sharded 2^%d MSM differs from the closed form" % log2n
                print("bench.py: ERROR: " + out["error"], file=sys.stderr, flush=True)
        return out
    finally:
        if ex is not None:
            ex.close()
        sm.close()


def config4_wave_leg(k16, torch, dist, rank, world, dev, wave, scale):
    """BASELINE config 4: a wave of 64 DISTINCT witnesses of a VALID synthetic key of the Keyless shape (tests/valid_key_builder.py:
    a trapdoor set-up, so its proofs verify), one prover per GPU, proof j on rank j mod N ("one proof per GPU per wave"), every
    proof through k16_prover_prove_mem with CSPRNG blinding; then ALL 64 are accepted by ONE k16_verify_batch on rank 0 with
    their own public inputs and rejected with their neighbours' -- the service's loop, prover_handler.rs:244-345."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import groth16_io as gio
    import valid_key_builder as vkb
    ctx = k16.Context(dev)
    zpath = "/tmp/k16_bench_wave_%s.zkey" % _tag(dist)
    meta = [None]
    t0 = time.time()
    if rank == 0:
        try:                                          # (a failure here travels in the broadcast: nobody is left waiting)
            key = vkb.build(lambda g, sc: ctx.synth_points_scalars(g, sc), max(int(1209229 * scale), 64), max(int(107487 * scale), 8),
                            max(int(26870 * scale), 8), seed=11)
            with open(zpath + ".part", "wb") as f:
                f.write(key["zkey"])
            os.replace(zpath + ".part", zpath)
            meta = [{k: key[k] for k in ("shape", "vk", "n_vars", "domain", "n_coefs")}]
            del key
        except Exception as e:
            meta = [{"error": repr(e)}]
    if dist is not None:
        dist.broadcast_object_list(meta, src=0)
    meta = meta[0]
    if "error" in meta:
        ctx.close()
        return {"error": "the valid key could not be built: %s" % meta["error"]} if rank == 0 else None
    t_key = time.time() - t0
    prover, others, V = None, [], None
    try:
        setup_err = None
        try:                                          # local set-up: no collective in here
            prover = k16.Prover(ctx, zpath)
            mine = [j for j in range(wave) if j % world == rank]
            t0 = time.time()
            wits = {j: vkb.fast_witness(meta["shape"], 5000 + j) for j in mine}
            t_wit = time.time() - t0
            for j in mine[:2]:
                prover.prove_mem(wits[j][0])          # warm-up (the clocks; the workspaces exist since create)
        except Exception as e:
            setup_err = repr(e)
        if not all_ranks_ok(torch, dist, setup_err is None):
            return {"error": "set-up failed on a rank (%s): leg skipped on all ranks" % (setup_err or "another rank")} if rank == 0 else None
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        out_js, lat = {}, []
        t_all = time.perf_counter()
        for j in mine:
            t1 = time.perf_counter()
            out_js[j] = prover.prove_mem(wits[j][0])
            lat.append((time.perf_counter() - t1) * 1e3)
        elapsed = time.perf_counter() - t_all
        inputs = {j: wits[j][1] for j in mine}
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=XDEV)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            allv = [None] * world
            dist.all_gather_object(allv, (out_js, inputs, lat))
            out_js, inputs, lat = {}, {}, []
            for a, b, c in allv:
                out_js.update(a)
                inputs.update(b)
                lat += c
        # one GPU, throughput mode: the same wave through TWO provers that share the device and ONE resident key
        # (k16_prover_create_shared -- what FullProver builds for K16_DEVICES=0,0)
        two = None
        if world == 1 and int(os.environ.get("K16_BENCH_PROVERS", "2")) > 1:
            import threading
            c2 = k16.Context(dev)
            t0 = time.time()
            p2 = k16.Prover(c2, zpath, share_key_of=prover)
            t_shared = time.time() - t0
            others = [(p2, c2)]
            for c in (ctx, c2):
                c.set_option(k16.OPT_SHARED_GPU, 1)
            p2.prove_mem(wits[0][0])
            jobs, lock, lat2, js2 = list(range(wave)), threading.Lock(), [], {}

            def worker(pv):
                while True:
                    with lock:
                        if not jobs:
                            return
                        j = jobs.pop(0)
                    t1 = time.perf_counter()
                    js2[j] = pv.prove_mem(wits[j][0])
                    lat2.append((time.perf_counter() - t1) * 1e3)

            t_all = time.perf_counter()
            th = [threading.Thread(target=worker, args=(pv,)) for pv in (prover, p2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            el2 = time.perf_counter() - t_all
            ctx.set_option(k16.OPT_SHARED_GPU, 0)
            two = {"provers": 2, "shared_resident_key": True, "second_prover_create_s": t_shared, "proofs_per_s": wave / el2,
                   "p50_ms": float(np.median(lat2)), "p99_ms": float(np.percentile(lat2, 99)), "_proofs": js2}
        out = None
        if rank == 0:
            V = k16.VerifyingKey(ctx, meta["vk"])
            order = list(range(wave))
            proofs = [gio.proof_from_json(out_js[j]) for j in order]
            ins = [inputs[j] for j in order]
            V.verify_batch(proofs[:2], ins[:2])               # warm-up of the verifier kernels
            t1 = time.perf_counter()
            ok = V.verify_batch(proofs, ins)
            t_verify = time.perf_counter() - t1
            wrong = V.verify_batch(proofs, [ins[(j + 1) % wave] for j in order])
            distinct = len(set(proofs)) == wave and len({tuple(x) for x in ins}) == wave
            out = {"workload": "wave of %d distinct witnesses, valid synthetic Keyless-shape key (nVars %d, N 2^%d, %d coefficients); "
                               "proof j on rank j mod N" % (wave, meta["n_vars"], int(np.log2(meta["domain"])), meta["n_coefs"]),
                   "entry": "k16_prover_prove_mem (CSPRNG blinding), one prover per GPU; k16_verify_batch of the whole wave on rank 0",
                   "ranks_seen": world, "proofs": wave, "proofs_per_s": wave / elapsed, "wave_s": elapsed,
                   "p50_ms": float(np.median(lat)), "p99_ms": float(np.percentile(lat, 99)),
                   "verify_wave_ms": t_verify * 1e3, "all_accepted": bool(all(ok)), "wrong_inputs_rejected": not any(wrong),
                   "distinct": bool(distinct), "proofs_per_s_proved_and_verified": wave / (elapsed + t_verify),
                   "setup_s": {"build_valid_key": t_key, "witnesses": t_wit},
                   "note": "a different key from the `proof` leg's: valid (its proofs verify), 3.7 M coefficients, and every bit wire has "
                           "non-zero B1 / B2 points (the proof leg's synthetic key has half of those rows (0,0)), so the B2 MSM is twice "
                           "the work and a proof ~0.8 ms longer; 64 distinct 43 MB witnesses, none of them cache-warm"}
            if two is not None:
                js2 = two.pop("_proofs")
                pr2 = [gio.proof_from_json(js2[j]) for j in order]
                two["all_accepted"] = bool(all(V.verify_batch(pr2, ins)))
                out["two_provers_one_gpu"] = two
                if not two["all_accepted"]:
                    out["error"] = "a proof of the two-prover wave was rejected"
            if not (all(ok) and not any(wrong) and distinct):
                out["error"] = "config 4 wave: a proof was rejected (or accepted with a foreign input)"
            if "error" in out:   # a SECONDARY leg: say so loudly, keep the line
                print("bench.py: ERROR: " + out["error"], file=sys.stderr, flush=True)
        return out
    finally:
        if V is not None:
            V.close()
        for pv, c in others:
            pv.close()
            c.close()
        if prover is not None:
            prover.close()
        ctx.close()
        if dist is not None:
            dist.barrier()          # nobody is still opening the key
        if rank == 0 and os.path.exists(zpath):
            os.unlink(zpath)


# ---------------------------------------------------------------------------------------------------- launch
def spawn_ranks(args):
    """--gpus N > 1 without a torchrun environment: start the N ranks as a child process.  Called before torch or the
    HIP library is imported -- a process that has touched the GPU must never be replaced or re-launched."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def cold_probe_child(args):
    """A FRESH process, no prewarm and no warm-up steps: the first MSM (workspace allocation, code objects, idle clocks) and
    the average of the first 20, pipelined exactly like the timed region.  What a service sees right after start-up; the
    headline figure is the steady state after bench.py's prewarm (`prewarm_steps`)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import k16
    ctx = k16.Context(0)
    ctx.set_option(k16.OPT_PIPELINED_MSM, 1)
    n = 1 << args.log2n
    d_bases = ctx.synth_points(k16.G1, 0, n)
    d_scalars = ctx.to_device(uniform_scalars(n, seed=0xD1B5))
    ctx.sync()
    t0 = time.perf_counter()
    ctx.set_lane(0)
    ctx.msm_enqueue(k16.G1, d_bases, d_scalars, n)
    ctx.msm_finish(k16.G1)
    first = (time.perf_counter() - t0) * 1e3
    depth, steps, lane = 4, 20, 1
    t0 = time.perf_counter()
    for k in range(min(depth - 1, steps)):
        ctx.set_lane(lane)
        lane = (lane + 1) % depth
        ctx.msm_enqueue(k16.G1, d_bases, d_scalars, n)
    for k in range(steps):
        if k + depth - 1 < steps:
            ctx.set_lane(lane)
            lane = (lane + 1) % depth
            ctx.msm_enqueue(k16.G1, d_bases, d_scalars, n)
        ctx.msm_finish(k16.G1)
    per = (time.perf_counter() - t0) * 1e3 / steps
    print(json.dumps({"first_step_ms": first, "ms_per_step_first20": per,
                      "note": "fresh process, no prewarm, no warm-up; steps 2..21 pipelined four deep like the timed region"}), flush=True)
    ctx.close()
    return 0


def warm_only_child(args):
    """A FRESH process doing exactly what the driver's flags say and nothing more: W warm-up steps, then K timed steps --
    no prewarm.  Reported beside the headline as `value_driver_warmup_only`: the headline's timed region
    starts after `prewarm_steps` more MSMs, when workspaces exist and the clocks have ramped up."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import k16
    ctx = k16.Context(0)
    ctx.set_option(k16.OPT_PIPELINED_MSM, 1)
    n = 1 << args.log2n
    d_bases = ctx.synth_points(k16.G1, 0, n)
    d_scalars = ctx.to_device(uniform_scalars(n, seed=0xD1B5))
    ctx.sync()
    depth = 4
    lane = [0]

    def run(steps):
        def enq():
            ctx.set_lane(lane[0])
            lane[0] = (lane[0] + 1) % depth
            ctx.msm_enqueue(k16.G1, d_bases, d_scalars, n)
        for k in range(min(depth - 1, steps)):
            enq()
        for k in range(steps):
            if k + depth - 1 < steps:
                enq()
            ctx.msm_finish(k16.G1)

    run(args.warmup)
    ctx.sync()
    t0 = time.perf_counter()
    run(args.steps)
    ctx.sync()
    el = time.perf_counter() - t0
    print(json.dumps({"value": n * args.steps / el, "ms_per_step": el / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                      "note": "fresh process, the driver's W warm-up steps and K timed steps only (no prewarm), one host thread"}),
          flush=True)
    ctx.close()
    return 0


def cold_probe(args, flag="--cold-probe"):
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), flag, "--log2n", str(args.log2n),
                              "--steps", str(args.steps), "--warmup", str(args.warmup)],
                             capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (out.stderr or out.stdout)[-300:]}
    except Exception as e:   # a report, never the measured path
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 0.2 s of GPU time: long enough to amortise pipeline fill / drain
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=LOG2N)
    ap.add_argument("--mode", choices=("weak", "strong"), default="weak")
    ap.add_argument("--total-log2n", type=int, default=26, help="strong mode: the whole MSM has 2^this points")
    ap.add_argument("--proofs", type=int, default=20, help="full Keyless-shape proofs per rank in the proof leg (0: skip it)")
    ap.add_argument("--proof-scale", type=float, default=1.0, help="shrink the synthetic circuit (1.0 = Keyless shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cold-probe", action="store_true", help=argparse.SUPPRESS)   # child of the main run, see cold_probe()
    ap.add_argument("--warm-only-probe", action="store_true", help=argparse.SUPPRESS)   # child: W warm-up + K steps, no prewarm
    args = ap.parse_args()
    if args.cold_probe:
        return cold_probe_child(args)
    if args.warm_only_probe:
        return warm_only_child(args)

    n_gpus = max(args.gpus, 1)
    if "WORLD_SIZE" not in os.environ and n_gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (n_gpus, world))

    # the cold figures come from a child process that runs (and ends) before this one touches the GPU
    cold = cold_probe(args) if (world == 1 and args.mode == "weak" and not os.environ.get("K16_BENCH_NO_COLD")) else None
    warm_only = cold_probe(args, "--warm-only-probe") if cold is not None else None

    # ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Four MSM lanes fill
    # them; with torch.distributed the RCCL stream would have to share one with a lane (measured at world size 1:
    # 538 M points/s with 4 queues, 598 M with 8).  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    if os.environ.get("K16_BENCH_CPUS"):   # experiments: pin the host thread (tools/numa_probe.py)
        cpus = set()
        for part in os.environ["K16_BENCH_CPUS"].split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus)

    import torch  # device plumbing + torch.distributed (RCCL); loaded first so one HIP runtime is shared
    import k16

    global SHARE_GPU, XDEV
    SHARE_GPU = bool(os.environ.get("K16_BENCH_SHARE_GPU"))
    XDEV = "cpu" if SHARE_GPU else "cuda"     # where the 128-byte exchange and the timing reductions live

    dist = None
    if world > 1 or os.environ.get("K16_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # K16_BENCH_SHARE_GPU=1 (test rig for the N > 1 code path on a one-GPU box): every rank uses GPU 0 and the exchange
        # runs over gloo (RCCL refuses two ranks on one device); the sharding, the fold, the closed-form check over all
        # ranks and the replica proof leg are the same code as with one GPU per rank
        if SHARE_GPU:
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    dev = local_rank if (world > 1 and not SHARE_GPU) else 0
    ctx = k16.Context(dev)  # raises without a GPU / library: there is no CPU fallback
    strong = args.mode == "strong"
    depth_default = "1" if strong else "4"
    # steps are pipelined over the MSM lanes: throughput tuning (include/k16.h, K16_OPT_PIPELINED_MSM) -- 16 slots per
    # lane in the weighted bucket sum, and consecutive accumulations fenced so that the HIP events time execution only
    if int(os.environ.get("K16_BENCH_DEPTH", depth_default)) > 1:
        ctx.set_option(k16.OPT_PIPELINED_MSM, 1)
    # K16_BENCH_GRAPHS=1: the ~50 launches of an MSM's sort and reduction replayed as two HIP graphs (K16_OPT_GRAPHS).
    # Off by default: on ROCm 7.2 a hipGraphLaunch of ~25 nodes costs more host time than the launches it replaces
    if os.environ.get("K16_BENCH_GRAPHS", "0") != "0":
        ctx.set_option(k16.OPT_GRAPHS, 1)

    import sharding

    # K16_BENCH_SHARDS=k (strong mode, one process): the MSM through k16_msm_sharded_* -- k shards over the visible devices
    # (round-robin; on a one-GPU box k contexts on device 0), scalars handed over in HOST memory every step
    shards_obj, shard_devices = None, []
    if strong and world == 1 and int(os.environ.get("K16_BENCH_SHARDS", "0")) > 0:
        if os.environ.get("K16_BENCH_PREPARED") or os.environ.get("K16_BENCH_FIXED_BASE"):
            sys.exit("bench.py: K16_BENCH_SHARDS does not combine with K16_BENCH_PREPARED / K16_BENCH_FIXED_BASE "
                     "(the shards hold their own prepared slices; the main context has no table)")
        ndev = max(1, k16.load().k16_device_count())
        shard_devices = [r % ndev for r in range(int(os.environ["K16_BENCH_SHARDS"]))]
    if strong:
        total = 1 << args.total_log2n
        lo, hi = sharding.shard_range(total, world, rank)
        n, start = hi - lo, lo
        scalars = fast_scalars(n, seed=0xD1B5 + rank)
    else:
        n, start = 1 << args.log2n, rank * (1 << args.log2n)
        scalars = uniform_scalars(n, seed=0xD1B5 + rank)
    # this rank's shard: bases (start + i + 1) * G, own scalars
    if shard_devices:
        shards_obj = k16.ShardedMsm(shard_devices, k16.G1, n)
        for r in range(shards_obj.count()):
            slo, shi = shards_obj.shard_range(r)
            sc = shards_obj.shard_ctx(r)
            d = sc.synth_points(k16.G1, slo, shi - slo)
            shards_obj.set_bases_device(r, d)
            d.free()
        d_bases, d_scalars = ctx.alloc(16), ctx.alloc(16)
    else:
        d_bases = ctx.synth_points(k16.G1, start, n)
        d_scalars = ctx.to_device(scalars)

    if os.environ.get("K16_BENCH_C"):           # experiments only: override the automatic window size
        ctx.set_window_bits(int(os.environ["K16_BENCH_C"]))

    lane = [0]

    # K16_BENCH_PREPARED=1 (not the headline): the point table converted once to the kernels' row layout, as the prover
    # does with the zkey's static tables (k16_msm_bases_prepare); the default passes the reference-format table every step
    prepared = ctx.bases_prepare(k16.G1, d_bases, n) if os.environ.get("K16_BENCH_PREPARED") else None
    # K16_BENCH_FIXED_BASE=1 (not the headline either): precomputed window tables for a static point table
    fixed_tab = ctx.fixed_base_prepare(k16.G1, d_bases, n)[0] if os.environ.get("K16_BENCH_FIXED_BASE") else None

    def enqueue():
        ctx.set_lane(lane[0])            # cycle the MSM lanes (stream + workspace): consecutive MSMs overlap
        lane[0] = (lane[0] + 1) % depth_cell[0]
        if fixed_tab is not None:
            ctx.msm_enqueue_fixed_base(k16.G1, fixed_tab, d_scalars, n)
        elif prepared is not None:
            ctx.msm_enqueue_prepared(k16.G1, prepared, d_scalars, n)
        else:
            ctx.msm_enqueue(k16.G1, d_bases, d_scalars, n)

    pending_x = []   # the previous step's exchange, still in flight

    # The library's own RCCL leg (k16_rank_comm_*: ncclAllGather issued by libk16.so, sharding.RankExchange) is the carrier in BOTH
    # modes -- what SCALE measures is then the path a C++ service has.  Weak mode keeps its overlap: k16_rank_comm_allgather_start
    # enqueues step k's exchange, _finish completes step k - 1's.  K16_BENCH_EXCHANGE=torch selects torch.distributed's
    # all_gather; a C leg that cannot start on every rank falls back to it LOUDLY (stderr + `exchange_note`).  On the one-GPU
    # test rig (K16_BENCH_SHARE_GPU) real RCCL refuses two ranks per device: the C leg runs there only over the test double
    # (K16_RCCL_LIB = tests/cpp/fake_rccl.cpp).
    c_exchange, c_exchange_note = make_rank_exchange(dist, ctx, sharding)

    c_in_flight = [0]

    def exchange(xyzz):
        if c_exchange is not None:
            if strong:
                return c_exchange.exchange_and_fold(k16.G1, xyzz)[0]
            c_exchange.start(k16.G1, xyzz)        # this step's gather runs under the next step's GPU work ...
            c_in_flight[0] += 1
            if c_in_flight[0] > 1:                # ... and the previous step's is completed and folded now
                c_in_flight[0] -= 1
                return c_exchange.finish()[0]
            return xyzz
        if dist is not None:
            # the path's one exchange: start this step's all_gather, complete the previous step's (it ran under the
            # GPU work enqueued in between); run() drains the last one inside the timed region
            pending_x.append(sharding.exchange_start(dist, k16.G1, xyzz, device=None if SHARE_GPU else "cuda"))
            if len(pending_x) > 1:
                xyzz, _ = sharding.exchange_finish(pending_x.pop(0))
        return xyzz

    def finish():
        xyzz, _ = ctx.msm_finish(k16.G1)   # waits for THIS MSM only, then conversion + Horner on the host
        return exchange(xyzz)

    depth_cell = [max(1, min(int(os.environ.get("K16_BENCH_DEPTH", depth_default)), 4))]
    chunked = n > (1 << 24)   # one device pass takes 2^24 points; above that k16_msm runs chunks on two lanes and folds them

    import threading
    if os.environ.get("K16_BENCH_SWITCH"):
        sys.setswitchinterval(float(os.environ["K16_BENCH_SWITCH"]))

    def run(steps):
        """steps complete MSMs, up to `depth` of them in flight.  The launches of MSM k+depth-1 are issued by a second host
        thread while this one waits for MSM k and combines its partial sums (the C entry points allow one enqueuing and
        one finishing thread): on a host whose HIP calls are slow the ~40 launches of an MSM then no longer sit between two
        waits.  K16_BENCH_THREADED=0: everything from this thread."""
        depth = depth_cell[0]
        res = None
        if shards_obj is not None:
            for k in range(steps):
                res = shards_obj.run(scalars)[0]
        elif chunked:
            for k in range(steps):
                res = exchange(ctx.msm_device(k16.G1, d_bases, d_scalars, n)[0])
        # One host path for N = 1 and N > 1 (a 1 -> 8 scaling curve must not mix a host-side change into "scaling"): the
        # producer thread below, also under torch.distributed -- the exchange is issued by THIS thread only, the producer
        # calls nothing but the C entry points (ctypes drops the interpreter lock around them).
        elif depth == 1 or os.environ.get("K16_BENCH_THREADED", "1") == "0":
            for k in range(min(depth - 1, steps)):
                enqueue()
            for k in range(steps):
                if k + depth - 1 < steps:
                    enqueue()
                res = finish()
        else:
            room, ready, failed = threading.Semaphore(depth), threading.Semaphore(0), []

            def producer():
                try:
                    for _ in range(steps):
                        room.acquire()
                        te = time.perf_counter()
                        enqueue()
                        if step_times is not None:
                            enq_times.append((time.perf_counter() - te) * 1e3)
                        ready.release()
                except BaseException as e:   # surface in the main thread instead of deadlocking it
                    failed.append(e)
                    for _ in range(steps):
                        ready.release()

            th = threading.Thread(target=producer)
            th.start()
            for k in range(steps):
                ready.acquire()
                if failed:
                    break
                res = finish()
                if step_times is not None:
                    step_times.append(time.perf_counter())
                room.release()
            th.join()
            if failed:
                raise failed[0]
        while pending_x:
            res, _ = sharding.exchange_finish(pending_x.pop(0))
        while c_in_flight[0]:                     # the last step's exchange, inside the timed region
            c_in_flight[0] -= 1
            res = c_exchange.finish()[0]
        return res

    step_times = [] if os.environ.get("K16_BENCH_TRACE") else None   # diagnostics: completion time of every step
    enq_times = []
    # Device warm-up, part of the set-up (reported as "prewarm_steps"; the W warm-up steps and the K timed steps follow
    # unchanged).  A fresh process needs ~40 MSMs before it runs at its steady rate: every lane's first MSM allocates its
    # workspaces (5-12 ms each), and the GPU's clocks ramp up over the first ~30 ms of load (per-step time 1.83 -> 1.65 ms,
    # gpurun_out traces in profiles/r02/); with W = 5 the driver's 20 timed steps would measure mostly that ramp.
    prewarm = int(os.environ.get("K16_BENCH_PREWARM", "48" if n <= (1 << 21) else "2"))   # large shards: seconds per MSM
    if prewarm:
        run(prewarm)
    if args.warmup:
        run(args.warmup)

    # the contexts whose kernels this run times: the main one, or (K16_BENCH_SHARDS) the shards' -- the main context runs no MSM then
    stat_ctxs = [ctx] if shards_obj is None else [shards_obj.shard_ctx(r) for r in range(shards_obj.count())]

    def stats_enable(v):
        for c in stat_ctxs:
            c.stats_enable(v)

    def stats_reset():
        for c in stat_ctxs:
            c.stats_reset()

    def stats_get(name):
        got = [c.stats_get(name) for c in stat_ctxs]
        return sum(g[0] for g in got), sum(g[1] for g in got)

    stats_enable(2)   # HIP events around the dominant kernel only (2 per MSM): the timed region stays lean
    stats_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    result = run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if step_times:
        ts = step_times[-args.steps:]
        print("[bench trace] per-step ms: " + " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip([t0] + ts[:-1], ts)),
              file=sys.stderr, flush=True)
        print("[bench trace] enqueue ms (all steps incl. warm-up): " + " ".join("%.2f" % v for v in enq_times),
              file=sys.stderr, flush=True)
    launches, acc_ms = stats_get("msm_accumulate")
    # host side of one step (C entry points only): launches, waiting for the GPU, conversion + Horner
    host_ms = {k: stats_get(k)[1] / max(stats_get(k)[0], 1)
               for k in ("host_enqueue", "host_finish_wait", "host_finish_combine", "host_enqueue_max")}
    # the same kernel without a neighbour on the GPU (one MSM at a time), for reference next to the live figure; this
    # untimed pass also times the other stages
    stats_enable(1)
    stats_reset()
    saved_depth = depth_cell[0]
    depth_cell[0] = 1
    lane[0] = 0
    run(1 if chunked else 3)
    depth_cell[0] = saved_depth
    iso_launches, iso_ms = stats_get("msm_accumulate")
    stage_ms = {k: stats_get(k)[1] / max(stats_get(k)[0], 1)
                for k in ("msm_sort", "msm_accumulate", "msm_fold", "msm_reduce")}
    stats_enable(0)

    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=XDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks_seen = dist.get_world_size()

    # untimed: the last step's (folded) result against the closed form  (sum over all shards of s_i * (start + i + 1)) * G
    k_mine = weighted_sum_mod_r(scalars, start)
    if dist is not None:
        ks = [None] * ranks_seen
        dist.all_gather_object(ks, k_mine)
        k_all = sum(ks) % R_MOD
    else:
        k_all = k_mine
    result_checked = None
    if rank == 0:
        _, got_aff = k16.points_sum(k16.G1, np.frombuffer(result, dtype=np.uint8).reshape(1, 128))
        result_checked = bool(got_aff == scalar_times_g(ctx, k16, k_all))
        if not result_checked and not os.environ.get("K16_BENCH_NOCHECK"):   # (probe runs produce wrong results on purpose)
            raise SystemExit("bench.py: MSM result differs from the closed form sum s_i (i+1) * G")

    # ---- secondary figure: the shape of Curve::multiMulByScalar's contract (curve.hpp:209-215) -- scalars arrive in HOST memory
    # (page-locked) every call, the result goes back to the host; bases stay resident (a prover's tables are static).  One MSM
    # at a time: upload, sort, accumulate, reduce, combine, no overlap between calls.  PCIe-inclusive, never `value`.
    host_leg = None
    if rank == 0 and not strong and not chunked and not os.environ.get("K16_BENCH_NO_HOST_LEG"):
        try:
            pinned = torch.from_numpy(scalars).pin_memory()
            hp = pinned.numpy()
            d_s2 = ctx.alloc(hp.nbytes)
            ctx.set_lane(0)
            reps = 10
            for k in range(2 + reps):
                if k == 2:
                    ctx.sync()
                    th0 = time.perf_counter()
                d_s2.upload(hp)
                ctx.msm_enqueue(k16.G1, d_bases, d_s2, n)
                xh, _ = ctx.msm_finish(k16.G1)
            dt = (time.perf_counter() - th0) / reps
            ok = bool(k16.points_sum(k16.G1, np.frombuffer(xh, dtype=np.uint8).reshape(1, 128))[1] ==
                      k16.points_sum(k16.G1, np.frombuffer(result, dtype=np.uint8).reshape(1, 128))[1]) if dist is None else None
            host_leg = {"points_per_s": n / dt, "ms_per_msm": dt * 1e3, "calls": reps, "same_result_as_timed_region": ok,
                        "note": "pinned host scalars uploaded (32 MB over PCIe) inside every call, one MSM at a time, result XYZZ on host"}
            d_s2.free()
        except Exception as e:   # a report, never the measured path
            host_leg = {"error": repr(e)}

    # ---- the proof leg (every rank proves; rank 0 reports)
    if c_exchange is not None:
        c_exchange.close()      # (the communicator lives on this context, which the proof leg replaces)
    if shards_obj is not None:
        shards_obj.close()
    d_bases.free()
    d_scalars.free()
    proof = None
    ctx.set_option(k16.OPT_PIPELINED_MSM, 0)   # the prover wants the latency-tuned defaults (include/k16.h)
    if args.proofs > 0:
        # the prover gets a context of its own, as in a service: the MSM bench's context has four lane streams, and a stream
        # the prover never uses still costs a proof ~0.4 ms (p50 7.9 vs 7.5 ms measured; ROCm multiplexes a process's streams
        # onto its hardware queues)
        if not os.environ.get("K16_BENCH_PROOF_SAME_CTX"):
            ctx.close()
            ctx = k16.Context(dev)
        proof = proof_leg(ctx, k16, torch, dist, rank, world, args.proofs,
                          check_with_oracle=(world == 1 and not args.no_cpu_baseline), scale=args.proof_scale)

    # ---- BASELINE configs 4 and 5 as secondary legs of the SAME line, after the timed region (round 6): ON by default so that
    # the driver's one command records them at every N; K16_BENCH_NO_CONFIG_LEGS=1 (or _NO_WAVE / _NO_2P26) skips them for quick
    # runs, K16_BENCH_2P26_LOG2N / K16_BENCH_WAVE_SCALE shrink them for the test suite
    legs = {}
    no_legs = bool(os.environ.get("K16_BENCH_NO_CONFIG_LEGS"))
    if not no_legs and not os.environ.get("K16_BENCH_NO_WAVE"):
        try:
            legs["config4_wave"] = config4_wave_leg(k16, torch, dist, rank, world, dev, int(os.environ.get("K16_BENCH_WAVE", "64")),
                                                    float(os.environ.get("K16_BENCH_WAVE_SCALE", "1.0")))
        except SystemExit:
            raise
        except Exception as e:
            if dist is not None:
                raise               # (a rank that left a leg early would hang the others in the next collective)
            legs["config4_wave"] = {"error": repr(e)}
    if not no_legs and not os.environ.get("K16_BENCH_NO_2P26"):
        lg = int(os.environ.get("K16_BENCH_2P26_LOG2N", "26"))
        try:
            legs["strong_2p%d" % lg] = strong_leg(k16, torch, dist, sharding, rank, world, dev, lg)
        except SystemExit:
            raise
        except Exception as e:
            if dist is not None:
                raise
            legs["strong_2p%d" % lg] = {"error": repr(e)}

    if rank == 0:
        total_points = float(n) * args.steps if not strong else float(1 << args.total_log2n) * args.steps
        if not strong:
            total_points *= world
        value = total_points / elapsed
        kern_s = (acc_ms / max(launches, 1)) * 1e-3
        # rows one launch of the dominant kernel covers: the whole MSM up to 2^24 rows, a 2^24-row chunk above, one PIECE of one
        # shard (k16_msm_sharded_*: 2^22 rows of host scalars per device pass) in the one-process sharded run
        pts_per_launch = min(n, 1 << 24) if shards_obj is None else min((n + len(shard_devices) - 1) // len(shard_devices), 1 << 22)
        achieved = (pts_per_launch * ALGO_BYTES_PER_POINT) / kern_s / 1e9 if kern_s > 0 else 0.0
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and not strong and args.log2n == 20:
            try:
                tj = json.load(open(tp))
                # the counters were collected for ONE build of the kernel: the file carries a digest of the sources that
                # make k_accumulate<Eng9>; a tree whose kernel sources differ reports no traffic figure instead of a stale one
                if tj.get("kernel_sources_sha16") == kernel_sources_sha16():
                    traffic = tj.get("msm_accumulate_hbm_bytes_per_launch")
                    traffic_src = "replayed from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                                  "command at commit %s, same kernel sources as this tree: digest %s; not measured in this run)" \
                                  % (tj.get("commit"), tj.get("kernel_sources_sha16"))
                else:
                    traffic_src = "profiles/pmc_traffic.json was collected for other kernel sources (digest %s, this tree %s): " \
                                  "no figure" % (tj.get("kernel_sources_sha16"), kernel_sources_sha16())
            except Exception:
                traffic = None
        iso = iso_ms / max(iso_launches, 1)
        out = {
            "metric": "BN254 G1 MSM points/s @2^%d (Groth16 prover hot path)" % (args.total_log2n if strong else args.log2n),
            "value": value,
            "unit": "points/s",
            "n_gpus": ranks_seen,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": prewarm,
            "value_driver_warmup_only": (warm_only or {}).get("value"),
            "driver_warmup_only": warm_only,
            "cold": cold,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u64 column accumulators over 9 x 29-bit limbs (256-bit Montgomery integers, v_mad_u64_u32)",
            "data": "synthetic: bases (i+1)*G generated on device, scalars uniform in [0,%s) (numpy seed 0xD1B5+rank)"
                    % ("2^253" if strong else "r"),
            "result_checked": result_checked,
            "config": {
                "workload": ("BN254 G1 Pippenger MSM, ONE MSM of 2^%d points sharded over the ranks, result XYZZ on host"
                             % args.total_log2n) if strong else
                            ("BN254 G1 Pippenger MSM, 2^%d random scalars/points per GPU, result XYZZ on host" % args.log2n),
                "points_per_gpu": n,
                "scalars": "device-resident, the SAME array every step, its 32 MB H2D outside the timed region (kernel microbench; "
                           "`host_scalars_leg` is the figure with pinned host scalars uploaded inside every step)",
                "bases": "fixed-base window tables (k16_msm_fixed_base_prepare)" if fixed_tab is not None
                         else "prepared once (k16_msm_bases_prepare)" if prepared is not None
                         else "reference format (Montgomery affine), converted inside every step",
                "sharding": ("contiguous shards (sharding.shard_range) + %s all_gather of 128-B partials + EC-add fold"
                             % ("libk16.so's k16_rank_comm_* (ncclAllGather) over the RCCL test double (test rig: all ranks on GPU 0)"
                                if (c_exchange is not None and SHARE_GPU) else
                                "gloo (test rig: all ranks on GPU 0)" if SHARE_GPU else
                                "RCCL through libk16.so's k16_rank_comm_* (ncclAllGather)" if c_exchange is not None else
                                "RCCL (torch.distributed)")) if dist is not None else
                            ("single GPU" if shards_obj is None else
                             "ONE process, %d shards (k16_msm_sharded_*: a context per shard on devices %s, host-side EC-add fold)"
                             % (len(shard_devices), shard_devices)),
                "exchange_note": c_exchange_note,
            },
            "roofline": {
                "kernel": "k_accumulate<Eng9> (bucket accumulation, XYZZ mixed adds)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms": kern_s * 1e3,
                "kernel_ms_isolated": iso,
                "frac_isolated": (pts_per_launch * ALGO_BYTES_PER_POINT) / (iso * 1e-3) / 1e9 / HBM_PEAK_GBS if iso > 0 else None,
                # the same launch priced in the unit that actually bounds it: 254-bit modular multiplications
                "alu": {
                    "unit": "G modmul/s",
                    "achieved": pts_per_launch * 16 * MODMUL_PER_PAIR / (iso * 1e-3) / 1e9 if iso > 0 else None,
                    "peak": MODMUL_PEAK_G,
                    "frac": pts_per_launch * 16 * MODMUL_PER_PAIR / (iso * 1e-3) / 1e9 / MODMUL_PEAK_G if iso > 0 else None,
                    "frac_step": pts_per_launch * 16 * MODMUL_PER_PAIR / (elapsed / args.steps) / 1e9 / MODMUL_PEAK_G / (world if not strong else 1),
                    # ... and against the hardware's multiply-add issue rate (31.9 T v_mad_u64_u32 lane-ops/s / 162 per product)
                    "peak_hw": MODMUL_PEAK_HW_G,
                    "frac_hw": pts_per_launch * 16 * MODMUL_PER_PAIR / (iso * 1e-3) / 1e9 / MODMUL_PEAK_HW_G if iso > 0 else None,
                    "frac_step_hw": pts_per_launch * 16 * MODMUL_PER_PAIR / (elapsed / args.steps) / 1e9 / MODMUL_PEAK_HW_G / (world if not strong else 1),
                    "frac_step_basis": "the same algorithmic multiplications over the whole STEP (ms_per_step: sort, accumulation, "
                                       "fold, weighted sum, host combine of one MSM per lane, four lanes pipelined), per GPU",
                    "basis": "algorithmic 10 modmul x n x 16 windows per launch / kernel_ms_isolated; peak = measured "
                             "v_mad_u64_u32-bound multiply rate of the radix-2^29 field (no MFMA path exists for 254-bit integers)",
                },
                "note": "kernel_ms is the live average inside the timed region (HIP events on the kernel's stream; up to "
                        "four MSMs share the GPU, and the other lanes' sort / reduction kernels run beside it at a higher "
                        "wave priority, so their instructions are issued inside this interval: +0.3 ms over isolated, "
                        "+2 % points/s over not prioritising them); kernel_ms_isolated is the same kernel with one MSM at a "
                        "time. Integer-multiply-issue bound in practice; see DESIGN.md (modmul-rate view)",
            },
            "stage_ms_isolated": stage_ms,
            "host_ms": host_ms,
            "host_scalars_leg": host_leg,
            "proof": proof,
        }
        out.update(legs)
        if world == 1 and not args.no_cpu_baseline:
            try:
                n_cpu = 1 << min(args.log2n, 20)
                out["cpu_baseline"] = cpu_baseline(n_cpu, uniform_scalars(n_cpu, seed=0xD1B5) if strong else scalars)
                if proof and proof.get("cpu_oracle"):
                    out["cpu_baseline"]["proof"] = proof.pop("cpu_oracle")
            except Exception as e:  # the baseline is a report, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "points/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
