"""Experiment: T host threads, each with its own k16 context pipelining D MSMs over its lanes, sharing one GPU.
Prints whole-GPU MSM throughput for the bench.py workload (2^20 points)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
import torch  # noqa: F401,E402  (same HIP runtime as bench.py)
import k16  # noqa: E402
from bench import uniform_scalars  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1 << 20
ctxs, bufs = [], []
for t in range(T):
    c = k16.Context(0)
    c.set_option(k16.OPT_PIPELINED_MSM, 1)
    d_b = c.synth_points(k16.G1, 0, n)
    d_s = c.to_device(uniform_scalars(n, 0xD1B5 + t))
    ctxs.append(c)
    bufs.append((d_b, d_s))


def run(t, steps):
    c = ctxs[t]
    d_b, d_s = bufs[t]
    lane = 0
    inflight = 0
    done = 0
    issued = 0
    while done < steps:
        while issued < steps and inflight < D:
            c.set_lane(lane)
            lane = (lane + 1) % D
            c.msm_enqueue(k16.G1, d_b, d_s, n)
            issued += 1
            inflight += 1
        c.msm_finish(k16.G1)
        inflight -= 1
        done += 1


for t in range(T):
    run(t, 3)
bar = threading.Barrier(T + 1)


def worker(t):
    bar.wait()
    run(t, STEPS)


ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for th in ths:
    th.start()
torch.cuda.synchronize()
bar.wait()
t0 = time.perf_counter()
for th in ths:
    th.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("threads=%d depth=%d : %.1f M points/s  (%.3f ms per MSM)" % (T, D, T * STEPS * n / dt / 1e6, dt / (T * STEPS) * 1e3))
