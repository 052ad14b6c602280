"""One RANK of the one-process-per-GPU exchange (include/k16.h: k16_rank_comm_*), started by tests/test_gpu_rank_comm.py as a
child process: several of these share GPU 0 and talk through the RCCL test double (tests/cpp/fake_rccl.cpp, K16_RCCL_LIB).
No torch here: the only thing between the ranks is libk16.so's own collective.

    python rank_comm_child.py <rank> <world> <id-file> <mode> [n] [seed]

mode  msm   rank r computes the partial MSM of its contiguous shard (sharding.shard_range) of a seeded G1 and a seeded G2
            problem, three rounds with different scalars, exchanges every partial with allgather_fold and prints the folded
            affine results (hex) -- the parent compares them with the oracle's MSM over ALL rows
      die   every rank but 0 leaves right after the communicator exists; rank 0 calls allgather_fold and must get
            K16_ERR_HIP within the bounded wait, and again at once on a second call
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "keyless-zk-proofs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    rank, world, idfile, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 3000
    seed = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    import numpy as np
    import k16
    import sharding
    import oracle_lib as ol          # input generation only (the parent does the checking)
    from gpu_common import np_scalars

    ctx = k16.Context(0)
    if rank == 0:
        uid = k16.RankComm.unique_id()
        with open(idfile + ".part", "wb") as fh:
            fh.write(uid)
        os.replace(idfile + ".part", idfile)
    else:
        t_end = time.time() + 120
        while not os.path.exists(idfile):
            if time.time() > t_end:
                raise SystemExit("rank %d: no communicator id after 120 s" % rank)
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    comm = k16.RankComm(ctx, rank, world, uid)
    out = {"rank": rank, "world": world}
    if mode == "die":
        if rank != 0:
            os._exit(0)              # no destroy, no collective: what a crashed rank looks like to the others
        t0 = time.time()
        errs = []
        for _ in range(2):
            try:
                comm.allgather_fold(k16.G1, bytes(128))
                errs.append(0)
            except k16.K16Error as e:
                errs.append(e.rc)
            errs.append(round(time.time() - t0, 3))
        out["errs"] = errs
        out["msg"] = (ctx.L.k16_last_error(ctx.h) or b"").decode()
        print(json.dumps(out), flush=True)
        comm.close()
        ctx.close()
        return
    res = []
    for group in (k16.G1, k16.G2):
        nn = n if group == k16.G1 else max(n // 4, 8)
        bases = ol.gen_points(group, 3, nn)
        lo, hi = sharding.shard_range(nn, world, rank)
        for rnd in range(3):
            scalars = np_scalars(seed + 17 * rnd + group, nn, "full256" if rnd != 1 else "witness")
            xyzz, _ = ctx.msm(group, bases[lo:hi], scalars[lo:hi])
            _, aff = comm.allgather_fold(group, xyzz)
            res.append(aff.hex())
    out["results"] = res
    print(json.dumps(out), flush=True)
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
