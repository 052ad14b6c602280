"""-m gpu : ONE host-thread pool per process, however many contexts and provers it holds.
The reference has one TBB arena per process (rust-rapidsnark/rapidsnark/src/multiexp.cpp:46) and one prover behind a
mutex (prover-service/src/prover_state.rs:21); a service that keeps several provers per GPU over eight GPUs must not get
12 worker threads per prover."""
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _threads_of_process():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("Threads:"):
                return int(line.split()[1])
    return -1


def test_eight_provers_share_one_host_pool_and_keep_the_throughput():
    """Eight contexts with a prover each on ONE GPU (K16_DEVICES=0,0,0,0,0,0,0,0 in FullProver terms), a circuit large
    enough for the compact witness upload (the pool's main user): the pool's size does not depend on the number of
    provers and stays within the CPUs the process may use; proofs are the same bytes from every prover; eight provers
    proving at once are not slower than four (the GPU is the limit from four on)."""
    import k16
    import bench
    usable = len(os.sched_getaffinity(0))
    L = k16.load()
    ctx0 = k16.Context(0)
    n_vars, N, n_coefs = 335897, 1 << 19, 2075000          # the Keyless shape at scale 1/4
    zk = bench.synth_zkey_bytes(ctx0, k16, n_vars, 1, N, n_coefs)
    zpath = "/tmp/k16_threads_test.zkey"
    with open(zpath, "wb") as f:
        f.write(zk)
    del zk
    r, s = bench._le32(12345 % bench.R_MOD), bench._le32(67890 % bench.R_MOD)
    wit = bench.synth_witness(n_vars, 42)
    provers = [k16.Prover(ctx0, zpath)]
    workers_after_one = L.k16_host_threads()
    os_threads_after_one = _threads_of_process()
    ref = provers[0].prove_mem(wit, r, s)
    provers += [k16.Prover(k16.Context(0), zpath) for _ in range(7)]
    for pv in provers:                      # what FullProver does for K16_DEVICES=0,0,...: same proofs, throughput tuning
        pv.ctx.set_option(k16.OPT_SHARED_GPU, 1)
    assert L.k16_host_threads() == workers_after_one            # one pool, created once
    assert 0 < workers_after_one + 1 <= max(usable, 2)
    # (HIP itself starts a few threads per context; the library's own contribution must not grow with the provers)
    assert _threads_of_process() - os_threads_after_one <= 7 * 6

    def run(group, proofs_each):
        out = [None] * len(group)

        def worker(i):
            for _ in range(proofs_each):
                out[i] = group[i].prove_mem(wit, r, s)
        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(group))]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        assert all(o == ref for o in out)
        return len(group) * proofs_each / dt

    run(provers, 2)                                             # warm every prover's workspaces
    four = max(run(provers[:4], 12) for _ in range(2))
    eight = max(run(provers, 6) for _ in range(2))
    print("proofs/s at 1/4 Keyless shape: four provers %.1f, eight provers %.1f, pool workers %d, usable CPUs %d"
          % (four, eight, workers_after_one, usable))
    assert eight >= 0.85 * four
    for p in provers:
        p.close()
