#!/bin/bash
# Collects every measurement DESIGN.md / README quote for this round on the GPU box into gpurun_out/$ROUND/ (copy the
# summaries into profiles/$ROUND/ afterwards; K16_COMMIT = the commit the tree was taken at, the GPU box has no .git).  Run through gpurun from the repository root.
set -u
O=gpurun_out/${ROUND:-r05}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. the driver's command, plain
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 bench.py --steps 100 --warmup 5 --proofs 0 --no-cpu-baseline > $O/bench_100steps.json 2>/dev/null
# 2. the same command under rocprofv3 --kernel-trace --stats (kernel averages the roofline must agree with)
K16_BENCH_NO_COLD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --proofs 0 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
# 3. PMC traffic, one counter per pass, one MSM at a time
K16_BENCH_NO_COLD=1 K16_BENCH_DEPTH=1 K16_BENCH_PREWARM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
K16_BENCH_NO_COLD=1 K16_BENCH_DEPTH=1 K16_BENCH_PREWARM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
# 4. instruction-rate microbenchmarks (the multiply peak the ALU roofline uses)
./tools/ubench > $O/ubench_instruction_rates.log 2>&1
# 5. proofs: latency / throughput / facade, and the kernel statistics of a proof
python3 tools/bench_proof.py --proofs 20 --facade 1 > $O/proof_keyless_shape.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/proof_stats -- python3 tools/bench_proof.py --proofs 10 > /dev/null 2> $O/proof_stats.err
# 6. the multi-GPU code path on this one GPU: RCCL exchange forced at world size 1, and the strong-scaling leg (one 2^26 MSM)
K16_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 40 --warmup 5 --proofs 0 --no-cpu-baseline > $O/bench_force_dist_world1.json 2> $O/bench_force_dist_world1.err
python3 bench.py --mode strong --total-log2n 26 --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > $O/bench_strong_2p26_1gpu.json 2> $O/bench_strong_2p26_1gpu.err
K16_BENCH_SHARE_GPU=1 K16_BENCH_PREWARM=2 python3 bench.py --gpus 2 --steps 4 --warmup 1 --log2n 20 --proofs 0 --no-cpu-baseline > $O/bench_2ranks_shared_gpu_weak.json 2> $O/bench_2ranks_shared_gpu_weak.err
K16_BENCH_SHARE_GPU=1 K16_BENCH_PREWARM=2 python3 bench.py --gpus 2 --mode strong --total-log2n 24 --steps 2 --warmup 1 --proofs 0 --no-cpu-baseline > $O/bench_2ranks_shared_gpu_strong.json 2> $O/bench_2ranks_shared_gpu_strong.err
# 6b. round 5: the C entry points of the sharded MSM on this one GPU -- the strong-mode bench through k16_rank_comm_* (world size 1,
#     RCCL) and through k16_msm_sharded_* (two contexts on device 0)
K16_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --mode strong --total-log2n 24 --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > $O/bench_strong_2p24_c_exchange_world1.json 2> $O/bench_strong_2p24_c_exchange_world1.err
K16_BENCH_SHARDS=2 python3 bench.py --mode strong --total-log2n 24 --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > $O/bench_strong_2p24_two_shards_one_process.json 2> $O/bench_strong_2p24_two_shards_one_process.err
# 7. verifier
python3 tools/bench_verify.py > $O/verify.json 2> $O/verify.err
# 8. BASELINE config 4 on one GPU: a wave of proofs of a valid synthetic key from a prover pool, one batched verification
python3 tools/config4_wave.py --provers 3 --wave 96 > $O/config4_wave_1gpu.json 2> $O/config4_wave_1gpu.err
# 9. kernel timeline of one proof, the isolated fixed-base H MSM, the NTT through the public entry point
rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 > /dev/null 2> $O/proof_timeline.err
python3 tools/proof_timeline.py /tmp/k16_tl 2 > $O/proof_timeline.txt 2>> $O/proof_timeline.err
python3 tools/fixed_base_timing.py 21 8 > $O/fixed_base_h_msm.log 2>&1
python3 tools/ntt_timing.py 21 30 > $O/ntt_2p21.log 2>&1
# 10. sort lab (per-kernel averages of the product's bucket sort), verifier trace, PMC write traffic of the sort
tools/lab/sortlab 20 16 0 uniform 20 check > $O/sortlab_2p20_c16.log 2>&1
tools/lab/sortlab 21 20 1 uniform 20 check > $O/sortlab_2p21_flat20.log 2>&1
tools/lab/prof.sh ${ROUND:-r05}s16 tools/lab/sortlab 20 16 0 uniform 20 > $O/sortlab_2p20_c16_kernels.txt 2>&1
tools/lab/prof.sh ${ROUND:-r05}sH tools/lab/sortlab 21 20 1 uniform 20 > $O/sortlab_2p21_flat20_kernels.txt 2>&1
tools/lab/pmc.sh ${ROUND:-r05}w16 WRITE_SIZE tools/lab/sortlab 20 16 0 uniform 3 > $O/sortlab_2p20_c16_WRITE_SIZE.txt 2>&1
tools/lab/pmc.sh ${ROUND:-r05}wH WRITE_SIZE tools/lab/sortlab 21 20 1 uniform 3 > $O/sortlab_2p21_flat20_WRITE_SIZE.txt 2>&1
K16_NO_STAGED_SORT=1 tools/lab/sortlab 21 20 1 uniform 20 check > $O/sortlab_2p21_flat20_round3_sort.log 2>&1
K16_NO_STAGED_SORT=1 tools/lab/sortlab 20 16 0 uniform 20 check > $O/sortlab_2p20_c16_round3_sort.log 2>&1
# 10a. round 5: instruction costs, gaps between the pipelined accumulations, provers per GPU
tools/lab/ubench2 > $O/ubench2_instruction_costs.log 2>&1
rm -rf /tmp/k16_msm_tl; K16_BENCH_NO_COLD=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_msm_tl -- python3 bench.py --steps 20 --warmup 5 --proofs 0 --no-cpu-baseline > /dev/null 2> $O/msm_pipeline_gaps.err
python3 tools/lab/msm_pipeline_gaps.py /tmp/k16_msm_tl > $O/msm_pipeline_gaps.txt 2>> $O/msm_pipeline_gaps.err
for p in 2 3 4; do K16_BENCH_PROVERS=$p K16_BENCH_NO_COLD=1 python3 bench.py --steps 5 --warmup 2 --proofs 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['proof']['throughput_mode']
print('provers $p: %.1f proofs/s  p50 %.2f ms   (one proof at a time: p50 %.2f ms)' % (t['proofs_per_s'], t.get('p50_ms',0), d['proof']['p50_ms']))"; done > $O/throughput_provers_sweep.log 2>&1
# 10b. round 4: counters of the bucket accumulation over fixed-base tables of three sizes, of the NTT
#      passes, VALU instructions per proof by kernel
tools/lab/gather_counters.sh > $O/pmc_fixed_base_accumulate_tlb_l2.txt 2>&1
tools/lab/ntt_counters.sh > $O/pmc_ntt_passes.txt 2>&1
tools/lab/valu_per_proof.sh ${ROUND:-r05} > $O/valu_instructions_per_proof.txt 2>&1
K16_VERIFY_COOP_TRACE=1 python3 tools/bench_verify.py > /dev/null 2> $O/verify_coop_trace.log   # (the trace build of the kernel is slower: not the numbers of record)
ls -la $O
# 11. summaries out of the raw rocprofv3 directories (what gets copied into profiles/$ROUND/)
cp "$(ls -S $(find $O/stats -name "*kernel_stats.csv") | head -1)" $O/bench_kernel_stats.csv   # (the largest: child processes write their own)
cp "$(ls -S $(find $O/proof_stats -name "*kernel_stats.csv") | head -1)" $O/proof_keyless_shape_kernel_stats.csv
python3 tools/pmc_kernel.py $O/pmc_fetch > $O/pmc_FETCH_SIZE_per_kernel.txt
python3 tools/pmc_kernel.py $O/pmc_write > $O/pmc_WRITE_SIZE_per_kernel.txt
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json ${K16_COMMIT:-unknown} > /dev/null
rm -rf $O/stats $O/proof_stats $O/pmc_fetch $O/pmc_write
