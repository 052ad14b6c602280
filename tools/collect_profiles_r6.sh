#!/bin/bash
# Round 6: collects what DESIGN.md / README quote for this round on the GPU box into gpurun_out/r06/ (copy the summaries into
# profiles/r06/ afterwards; K16_COMMIT = the commit the tree was taken at, the GPU box has no .git).  Run through gpurun from the
# repository root.  (tools/collect_profiles.sh is round 5's, with the kernel-level sweeps this round did not touch.)
set -u
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. the driver's command, plain (with the config 4 / 5 legs), and a 100-step run of the headline
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
K16_BENCH_NO_CONFIG_LEGS=1 python3 bench.py --steps 100 --warmup 5 --proofs 0 --no-cpu-baseline > $O/bench_100steps.json 2>/dev/null
# 2. the same command under rocprofv3 --kernel-trace --stats (kernel averages the roofline must agree with)
K16_BENCH_NO_COLD=1 K16_BENCH_NO_CONFIG_LEGS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --proofs 0 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
# 3. PMC traffic, one counter per pass, one MSM at a time
K16_BENCH_NO_COLD=1 K16_BENCH_NO_CONFIG_LEGS=1 K16_BENCH_DEPTH=1 K16_BENCH_PREWARM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
K16_BENCH_NO_COLD=1 K16_BENCH_NO_CONFIG_LEGS=1 K16_BENCH_DEPTH=1 K16_BENCH_PREWARM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --proofs 0 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
# 4. proofs: latency / throughput / facade, the kernel statistics and the kernel timeline of a proof, the NTT alone
python3 tools/bench_proof.py --proofs 30 --facade 1 --concurrent 2 --random-rs > $O/proof_keyless_shape.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/proof_stats -- python3 tools/bench_proof.py --proofs 10 > /dev/null 2> $O/proof_stats.err
rm -rf /tmp/k16_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tl -- python3 tools/bench_proof.py --proofs 6 --no-stats > /dev/null 2> $O/proof_timeline.err
python3 tools/proof_timeline.py /tmp/k16_tl 3 > $O/proof_timeline.txt 2>> $O/proof_timeline.err
python3 tools/ntt_timing.py 21 30 > $O/ntt_2p21.log 2>&1
# 5. the multi-rank code paths on this one GPU: both modes of bench.py on two ranks that share GPU 0, the exchange through the
#    library's RCCL leg over the test double (tests/cpp/fake_rccl.cpp), and the config 4 / 5 legs on two ranks at reduced sizes
mkdir -p /tmp/k16_fake && g++ -std=c++17 -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/fake_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib -o /tmp/k16_fake/librccl.so.1
K16_RCCL_LIB=/tmp/k16_fake/librccl.so.1 K16_BENCH_SHARE_GPU=1 K16_BENCH_PREWARM=2 K16_BENCH_NO_CONFIG_LEGS=1 python3 bench.py --gpus 2 --steps 6 --warmup 2 --log2n 20 --proofs 0 --no-cpu-baseline > $O/bench_2ranks_shared_gpu_weak_c_exchange.json 2> $O/bench_2ranks_weak.err
K16_RCCL_LIB=/tmp/k16_fake/librccl.so.1 K16_BENCH_SHARE_GPU=1 K16_BENCH_PREWARM=2 K16_BENCH_NO_CONFIG_LEGS=1 python3 bench.py --gpus 2 --mode strong --total-log2n 24 --steps 2 --warmup 1 --proofs 0 --no-cpu-baseline > $O/bench_2ranks_shared_gpu_strong_c_exchange.json 2> $O/bench_2ranks_strong.err
K16_RCCL_LIB=/tmp/k16_fake/librccl.so.1 K16_BENCH_SHARE_GPU=1 K16_BENCH_PREWARM=2 K16_BENCH_WAVE_SCALE=0.25 K16_BENCH_2P26_LOG2N=24 python3 bench.py --gpus 2 --steps 4 --warmup 1 --log2n 18 --proofs 0 --no-cpu-baseline > $O/bench_2ranks_shared_gpu_config_legs.json 2> $O/bench_2ranks_legs.err
# 6. summaries out of the raw rocprofv3 directories (what gets copied into profiles/r06/)
cp "$(ls -S $(find $O/stats -name "*kernel_stats.csv") | head -1)" $O/bench_kernel_stats.csv
cp "$(ls -S $(find $O/proof_stats -name "*kernel_stats.csv") | head -1)" $O/proof_keyless_shape_kernel_stats.csv
python3 tools/pmc_kernel.py $O/pmc_fetch > $O/pmc_FETCH_SIZE_per_kernel.txt
python3 tools/pmc_kernel.py $O/pmc_write > $O/pmc_WRITE_SIZE_per_kernel.txt
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json ${K16_COMMIT:-unknown} > /dev/null
rm -rf $O/stats $O/proof_stats $O/pmc_fetch $O/pmc_write
ls -la $O
