#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python3 tools/prove_soak.py 6000 2 > $O/prove_soak_keyless_shape_12000_final.log 2>&1; tail -1 $O/prove_soak_keyless_shape_12000_final.log
python3 tools/prove_fuzz.py 1000 7 > $O/prove_fuzz_1000_seed7_final.json 2>&1; tail -c 250 $O/prove_fuzz_1000_seed7_final.json; echo
python3 tools/msm_fuzz.py 1500 7 > $O/msm_fuzz_1500_seed7_final.json 2>&1; tail -c 250 $O/msm_fuzz_1500_seed7_final.json; echo
python3 tools/classes_fuzz.py 150 7 > $O/classes_fuzz_150_seed7_final.json 2>&1; tail -c 250 $O/classes_fuzz_150_seed7_final.json; echo
python3 tools/ntt_soak.py 21 200 > $O/ntt_soak_2p21_200_final.json 2>&1; tail -c 250 $O/ntt_soak_2p21_200_final.json; echo
python3 tools/msm_soak.py 20 100 > $O/msm_soak_2p20_100_final.json 2>&1; tail -c 250 $O/msm_soak_2p20_100_final.json; echo
