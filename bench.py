#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Groth16 hot path.

Workload (BASELINE.json configs[1]): BN254 G1 Pippenger MSM, 2^20 points, uniform random scalars in
[0, r), bases (i+1)*G -- all resident in HBM before the timed region.  One "step" = one complete MSM
(digits -> bucket sort -> bucket accumulation -> weighted bucket reduction -> Horner combine ->
XYZZ result on the host).  Steps are software-pipelined four deep over the context's MSM lanes (stream +
workspace each): the kernels of steps k+1 .. k+3 are enqueued before step k is finished, so the latency-bound
fold / reduction stages of one MSM overlap the next one's sort and accumulation and the 0.25 ms host tail is
hidden; the timed region still contains exactly K complete MSMs, each fully reduced to one point.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): the MSM shards naturally,
so every rank owns an independent 2^20-point shard of an N*2^20-point MSM (weak scaling), and each
step ends with the path's one real exchange: an RCCL all_gather of the 128-byte per-shard partial
results, folded on every rank with an EC add (RCCL has no EC-add reduction op).

The JSON line also carries
  roofline      -- for the dominant kernel (bucket accumulation): algorithmic bytes / measured launch time
  cpu_baseline  -- the CPU oracle (oracle/, a port of the reference algorithm) timed on this host on a
                   bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))

LOG2N = 20
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ALGO_BYTES_PER_POINT = 64 + 32  # G1 MSM = n x (64 B affine point + 32 B scalar)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
MODMUL_PER_PAIR = 10            # SURVEY 8(d) secondary figure: one mixed add (8M + 2S) per (point, window) pair
MODMUL_PEAK_G = 174.3           # measured on MI355X: radix-2^29 Montgomery multiply, all CUs (profiles/r01/ubench_*.log)


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


def cpu_baseline(n_sample, scalars):
    """Times the CPU oracle (port of the reference's ParallelMultiexp) on the first n_sample points."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol  # the checker, used here only as the reported CPU baseline

    threads = min(os.cpu_count() or 1, 16)  # the port parallelises over the 16 windows
    bases = ol.gen_points(0, 0, n_sample)
    sc = np.ascontiguousarray(scalars[:n_sample])
    ol.msm(0, bases[:4096], sc[:4096], nthreads=threads)  # warm-up
    t0 = time.time()
    ol.msm(0, bases, sc, nthreads=threads)
    dt = time.time() - t0
    return {
        "value": n_sample / dt,
        "unit": "points/s",
        "cores": threads,
        "kind": "port",
        "sample": "first 2^%d points of the same workload, %.1f s wall, oracle/bn254_ref.c (gcc -O2, OpenMP over windows)"
                  % (int(np.log2(n_sample)), dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 0.2 s of GPU time: long enough to amortise pipeline fill / drain
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=LOG2N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(args.gpus, 1)

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

    dist = None
    if world > 1 or os.environ.get("K16_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    dev = local_rank if world > 1 else 0
    ctx = k16.Context(dev)  # raises without a GPU / library: there is no CPU fallback
    # steps are pipelined over the MSM lanes: throughput tuning (include/k16.h, K16_OPT_PIPELINED_MSM) -- 16 slots per
    # lane in the weighted bucket sum, and consecutive accumulations fenced so that the HIP events time execution only
    if int(os.environ.get("K16_BENCH_DEPTH", "4")) > 1:
        ctx.set_option(k16.OPT_PIPELINED_MSM, 1)
    # K16_BENCH_GRAPHS=1: the ~50 launches of an MSM's sort and reduction replayed as two HIP graphs (K16_OPT_GRAPHS).
    # Off by default: on ROCm 7.2 a hipGraphLaunch of ~25 nodes costs more host time than the launches it replaces
    # (host_enqueue 0.14 -> 0.45 ms per MSM, 566 -> 470 M points/s)
    if os.environ.get("K16_BENCH_GRAPHS", "0") != "0":
        ctx.set_option(k16.OPT_GRAPHS, 1)

    n = 1 << args.log2n
    # this rank's shard of the (world * n)-point MSM: bases (rank*n + i + 1) * G, own scalars
    d_bases = ctx.synth_points(k16.G1, rank * n, n)
    scalars = uniform_scalars(n, seed=0xD1B5 + rank)
    d_scalars = ctx.to_device(scalars)

    if os.environ.get("K16_BENCH_C"):           # experiments only: override the automatic window size
        ctx.set_window_bits(int(os.environ["K16_BENCH_C"]))

    import sharding

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

    def finish():
        xyzz, _ = ctx.msm_finish(k16.G1)   # waits for THIS MSM only, then conversion + Horner on the host
        if dist is not None:
            # the path's one exchange: start this step's all_gather, complete the previous step's (it ran under the
            # GPU work enqueued in between); run() drains the last one inside the timed region
            pending_x.append(sharding.exchange_start(dist, k16.G1, xyzz, device="cuda"))
            if len(pending_x) > 1:
                xyzz, _ = sharding.exchange_finish(pending_x.pop(0))
        return xyzz

    depth_cell = [max(1, min(int(os.environ.get("K16_BENCH_DEPTH", "4")), 4))]

    import threading

    def run(steps):
        """steps complete MSMs, up to `depth` of them in flight.  The launches of MSM k+depth-1 are issued by a second host
        thread while this one waits for MSM k and combines its partial sums (the C entry points allow one enqueuing and
        one finishing thread): on a host whose HIP calls are slow the ~40 launches of an MSM then no longer sit between two
        waits.  K16_BENCH_THREADED=0: everything from this thread."""
        depth = depth_cell[0]
        res = None
        # (not with torch.distributed: its Python-side calls and the producer then fight over the interpreter lock)
        if depth == 1 or os.environ.get("K16_BENCH_THREADED", "1" if dist is None else "0") == "0":
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
                        enqueue()
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
                room.release()
            th.join()
            if failed:
                raise failed[0]
        while pending_x:
            res, _ = sharding.exchange_finish(pending_x.pop(0))
        return res

    if args.warmup:
        run(args.warmup)

    ctx.stats_enable(2)   # HIP events around the dominant kernel only (2 per MSM): the timed region stays lean
    ctx.stats_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    result = run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    launches, acc_ms = ctx.stats_get("msm_accumulate")
    # host side of one step (C entry points only): launches, waiting for the GPU, conversion + Horner
    host_ms = {k: ctx.stats_get(k)[1] / max(ctx.stats_get(k)[0], 1)
               for k in ("host_enqueue", "host_finish_wait", "host_finish_combine")}
    # the same kernel without a neighbour on the GPU (one MSM at a time), for reference next to the live figure; this
    # untimed pass also times the other stages
    ctx.stats_enable(1)
    ctx.stats_reset()
    saved_depth = depth_cell[0]
    depth_cell[0] = 1
    lane[0] = 0
    run(3)
    depth_cell[0] = saved_depth
    iso_launches, iso_ms = ctx.stats_get("msm_accumulate")
    stage_ms = {k: ctx.stats_get(k)[1] / max(ctx.stats_get(k)[0], 1)
                for k in ("msm_sort", "msm_accumulate", "msm_fold", "msm_reduce")}
    ctx.stats_enable(0)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_points = float(n) * world * args.steps
        value = total_points / elapsed
        kern_s = (acc_ms / max(launches, 1)) * 1e-3
        achieved = (n * ALGO_BYTES_PER_POINT) / kern_s / 1e9 if kern_s > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("msm_accumulate_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "BN254 G1 MSM points/s @2^%d (Groth16 prover hot path)" % args.log2n,
            "value": value,
            "unit": "points/s",
            "n_gpus": world if world > 1 else n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 column accumulators over 9 x 29-bit limbs (256-bit Montgomery integers, v_mad_u64_u32)",
            "data": "synthetic: bases (i+1)*G generated on device, scalars uniform in [0,r) (numpy seed 0xD1B5+rank)",
            "config": {
                "workload": "BN254 G1 Pippenger MSM, 2^%d random scalars/points per GPU, result XYZZ on host"
                            % args.log2n,
                "points_per_gpu": n,
                "bases": "fixed-base window tables (k16_msm_fixed_base_prepare)" if fixed_tab is not None
                         else "prepared once (k16_msm_bases_prepare)" if prepared is not None
                         else "reference format (Montgomery affine), converted inside every step",
                "sharding": "independent contiguous shards + RCCL all_gather of 128-B partials" if world > 1
                            else "single GPU",
            },
            "roofline": {
                "kernel": "k_accumulate<Eng9> (bucket accumulation, XYZZ mixed adds)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms": kern_s * 1e3,
                "kernel_ms_isolated": iso_ms / max(iso_launches, 1),
                "frac_isolated": (n * ALGO_BYTES_PER_POINT) / (iso_ms / max(iso_launches, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                 if iso_ms > 0 else None,
                # the same launch priced in the unit that actually bounds it: 254-bit modular multiplications
                "alu": {
                    "unit": "G modmul/s",
                    "achieved": n * 16 * MODMUL_PER_PAIR / (iso_ms / max(iso_launches, 1) * 1e-3) / 1e9 if iso_ms > 0 else None,
                    "peak": MODMUL_PEAK_G,
                    "frac": n * 16 * MODMUL_PER_PAIR / (iso_ms / max(iso_launches, 1) * 1e-3) / 1e9 / MODMUL_PEAK_G
                            if iso_ms > 0 else None,
                    "basis": "algorithmic 10 modmul x n x 16 windows per launch / kernel_ms_isolated; peak = measured "
                             "v_mad_u64_u32-bound multiply rate of the radix-2^29 field (no MFMA path exists for 254-bit integers)",
                },
                "note": "kernel_ms is the live average inside the timed region (HIP events on the kernel's stream; up to "
                        "three MSMs share the GPU, so the other lanes' sort / reduction kernels run beside it); "
                        "kernel_ms_isolated is the same kernel with one MSM at a time. Integer-multiply-issue bound in "
                        "practice; see DESIGN.md (modmul-rate view)",
            },
            "stage_ms_isolated": stage_ms,
            "host_ms": host_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(1 << min(args.log2n, 20), scalars)
            except Exception as e:  # the baseline is a report, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "points/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
