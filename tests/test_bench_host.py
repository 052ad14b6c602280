"""Host logic of bench.py that must be right without a GPU: the exact closed-form scalar of the MSM check, the
strong-scaling shard arithmetic, and the launch behaviour of `--gpus N` (`bench.py --gpus 2` must start two
ranks or fail, never print n_gpus = 2 from one process)."""
import os
import subprocess
import sys

import numpy as np

import pymodel as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "keyless-zk-proofs_amd"))
import bench  # noqa: E402
import sharding  # noqa: E402


def _ref_sum(sc, start):
    return sum(int.from_bytes(bytes(sc[i]), "little") * (start + i + 1) for i in range(sc.shape[0])) % pm.R


def test_weighted_sum_is_exact():
    sc = bench.uniform_scalars(3000, 7)
    assert all(int.from_bytes(bytes(r), "little") < pm.R for r in sc)
    assert bench.weighted_sum_mod_r(sc, 0) == _ref_sum(sc, 0)
    fs = bench.fast_scalars(5000, 3)
    assert all(int.from_bytes(bytes(r), "little") < (1 << 253) for r in fs)
    # start offsets of the last shard of a 2^26-point MSM over 8 ranks: the weight needs its third 14-bit part
    assert bench.weighted_sum_mod_r(fs, 7 * (1 << 23)) == _ref_sum(fs, 7 * (1 << 23))
    # chunk boundary of the 2^20-row blocks
    big = bench.fast_scalars((1 << 20) + 17, 5)
    part = bench.weighted_sum_mod_r(big[:1 << 20], 10) + bench.weighted_sum_mod_r(big[1 << 20:], 10 + (1 << 20))
    assert bench.weighted_sum_mod_r(big, 10) == part % pm.R


def test_strong_mode_shards_cover_the_msm_and_their_closed_forms_add_up():
    total, world = (1 << 16) + 5, 8
    sc = bench.fast_scalars(total, 11)
    ks, cover = 0, []
    for rank in range(world):
        lo, hi = sharding.shard_range(total, world, rank)
        cover.append((lo, hi))
        ks += bench.weighted_sum_mod_r(sc[lo:hi], lo)
    assert cover[0][0] == 0 and cover[-1][1] == total and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert ks % pm.R == bench.weighted_sum_mod_r(sc, 0)


def test_gpus_2_starts_ranks_or_fails_never_one_process():
    """Without GPUs the two ranks die at context creation: a non-zero exit code and no JSON line claiming n_gpus = 2."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--proofs", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, K16_BENCH_PREWARM="0"))
    has_gpu = out.returncode == 0
    if has_gpu:          # a real multi-GPU box: then two ranks must have been seen
        import json
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        assert json.loads(line)["ranks_seen"] == 2
    else:
        assert '"n_gpus": 2' not in out.stdout
        assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr or "No HIP GPUs" in out.stderr
    # a launcher that provides a different world size than --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--proofs", "0"],
                         capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
