"""-m gpu : the N > 1 path of bench.py, run every round on whatever box there is.

A one-GPU box cannot give two ranks a GPU each, and RCCL refuses two ranks on one device, so the two-rank runs here use
bench.py's test rig K16_BENCH_SHARE_GPU=1: both ranks compute on GPU 0 and the 128-byte exchange goes over gloo.  Everything
else is the production code path: bench.py spawning its ranks as a child process, sharding.shard_range, the per-rank MSM,
exchange_start / exchange_finish, the EC-add fold, the closed-form check over BOTH shards.  With two or more GPUs the same
commands also run over RCCL, one rank per GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    env.setdefault("K16_BENCH_NO_CONFIG_LEGS", "1")     # (the config 4 / 5 legs have tests of their own below)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("mode_args, scaling", [(["--log2n", "16"], "weak"),
                                                (["--mode", "strong", "--total-log2n", "20"], "strong")])
def test_bench_two_ranks_sharing_the_gpu(mode_args, scaling):
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--proofs", "0", "--no-cpu-baseline"] + mode_args,
               {"K16_BENCH_SHARE_GPU": "1", "K16_BENCH_PREWARM": "2"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["result_checked"] is True          # folded result of BOTH shards == (sum_i s_i (i+1)) G
    assert d["scaling"] == scaling and d["value"] > 0
    assert "gloo" in d["config"]["sharding"]


def test_bench_replica_proofs_on_two_ranks():
    """BASELINE config 4's replica mode on two ranks (one prover per rank, no collective), at a reduced circuit size so
    that two ranks fit one GPU's time budget: proofs/s is the sum over ranks."""
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "14", "--proofs", "3", "--proof-scale", "0.02",
                "--no-cpu-baseline"], {"K16_BENCH_SHARE_GPU": "1", "K16_BENCH_PREWARM": "2", "K16_BENCH_PROVERS": "1"})
    assert d["ranks_seen"] == 2 and d["result_checked"] is True
    assert d["proof"]["proofs"] == 6 and d["proof"]["proofs_per_s"] > 0
    assert d["proof"]["parallelism"].startswith("replicas")


def test_bench_one_rank_per_gpu_over_rccl():
    """With two or more GPUs: the same command over RCCL, one rank per GPU (skipped on the one-GPU boxes of the pool)."""
    import k16
    n = k16.load().k16_device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs; this box has %d" % n)
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--log2n", "18", "--proofs", "0", "--no-cpu-baseline"], {})
    assert d["ranks_seen"] == 2 and d["result_checked"] is True and "RCCL" in d["config"]["sharding"]


def test_bench_strong_mode_exchanges_through_the_c_entry_points():
    """BASELINE config 5's shape on one rank with the distributed path forced: the exchange is libk16.so's own RCCL leg
    (k16_rank_comm_create / k16_rank_comm_allgather_fold -- ncclCommInitRank + ncclAllGather issued by the library), world
    size 1 on the one-GPU boxes; the folded result is checked against the closed form."""
    d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--mode", "strong", "--total-log2n", "20", "--proofs", "0",
                "--no-cpu-baseline"], {"K16_BENCH_FORCE_DIST": "1", "K16_BENCH_PREWARM": "2"})
    assert d["result_checked"] is True and d["scaling"] == "strong"
    assert "k16_rank_comm" in d["config"]["sharding"], d["config"]


def test_bench_strong_mode_two_ranks_over_the_c_exchange():
    """Two GPUs: one rank per GPU, the 2^22-point MSM sharded, exchange through k16_rank_comm_* (skipped on one-GPU boxes)."""
    import k16
    n = k16.load().k16_device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs; this box has %d" % n)
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "strong", "--total-log2n", "22", "--proofs", "0",
                "--no-cpu-baseline"], {"K16_BENCH_PREWARM": "2"})
    assert d["ranks_seen"] == 2 and d["result_checked"] is True and "k16_rank_comm" in d["config"]["sharding"]


def test_bench_one_process_sharded_msm():
    """k16_msm_sharded_* under bench.py: ONE process, the strong-mode MSM cut into shards over the visible devices (two
    contexts on device 0 on a one-GPU box), scalars handed over in host memory, host-side fold, closed form checked."""
    d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--mode", "strong", "--total-log2n", "22", "--proofs", "0",
                "--no-cpu-baseline"], {"K16_BENCH_SHARDS": "2", "K16_BENCH_PREWARM": "1"})
    assert d["result_checked"] is True and "k16_msm_sharded" in d["config"]["sharding"]


def test_bench_config_legs_on_two_ranks_sharing_the_gpu():
    """The two secondary legs bench.py adds to its line at every N (BASELINE config 4: the 64-proof wave, proof j on rank
    j mod N, one k16_verify_batch of all of them on rank 0; config 5: ONE MSM cut over the ranks, partials exchanged and
    folded, closed form over all ranks' rows) on two ranks of the one-GPU test rig, at reduced sizes (a 2 % circuit, 2^18
    points) so that two ranks fit the box's time budget.  The full sizes run at N = 1 in the driver's bench and in
    test_config5_2p26_eight_shards_closed_form / test_config4_wave_of_64_distinct_proofs_one_verification_batch."""
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "14", "--proofs", "0", "--no-cpu-baseline"],
               {"K16_BENCH_SHARE_GPU": "1", "K16_BENCH_PREWARM": "1", "K16_BENCH_NO_CONFIG_LEGS": "", "K16_BENCH_WAVE_SCALE": "0.02",
                "K16_BENCH_2P26_LOG2N": "18", "K16_BENCH_WAVE": "16"})
    w, s5 = d["config4_wave"], d["strong_2p18"]
    assert w["ranks_seen"] == 2 and w["proofs"] == 16 and w["all_accepted"] and w["wrong_inputs_rejected"] and w["distinct"]
    assert s5["ranks_seen"] == 2 and s5["shards"] == 2 and s5["result_checked"] is True
    assert s5["host_scalars"]["points_per_s"] > 0 and s5["device_scalars"]["points_per_s"] > 0


def test_bench_config_legs_single_rank_small():
    """N = 1 form of the same legs (eight shards in one process; two provers sharing one resident key), reduced sizes."""
    d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--log2n", "14", "--proofs", "0", "--no-cpu-baseline"],
               {"K16_BENCH_PREWARM": "1", "K16_BENCH_NO_CONFIG_LEGS": "", "K16_BENCH_WAVE_SCALE": "0.06", "K16_BENCH_2P26_LOG2N": "20",
                "K16_BENCH_WAVE": "12", "K16_BENCH_NO_COLD": "1"})
    w, s5 = d["config4_wave"], d["strong_2p20"]
    assert w["all_accepted"] and w["two_provers_one_gpu"]["all_accepted"] and w["two_provers_one_gpu"]["shared_resident_key"]
    assert s5["shards"] == 8 and s5["result_checked"] is True and "k16_msm_sharded_run: 8 shards" in s5["entry"]
