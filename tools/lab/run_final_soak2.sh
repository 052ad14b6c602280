#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python3 tools/prove_soak.py 15000 2 > $O/prove_soak_keyless_shape_30000_final.log 2>&1; tail -1 $O/prove_soak_keyless_shape_30000_final.log
python3 tools/msm_fuzz.py 3000 13 > $O/msm_fuzz_3000_seed13_final.json 2>&1; tail -c 250 $O/msm_fuzz_3000_seed13_final.json; echo
python3 tools/classes_fuzz.py 300 13 > $O/classes_fuzz_300_seed13_final.json 2>&1; tail -c 250 $O/classes_fuzz_300_seed13_final.json; echo
python3 tools/ntt_fuzz.py 1000 13 > $O/ntt_fuzz_1000_seed13_final.json 2>&1; tail -c 250 $O/ntt_fuzz_1000_seed13_final.json; echo
python3 tools/pairing_fuzz.py 200 13 > $O/pairing_fuzz_200_seed13_final.json 2>&1; tail -c 250 $O/pairing_fuzz_200_seed13_final.json; echo
python3 tools/msm_soak.py 18 60 1 > $O/msm_soak_g2_2p18_60_final.json 2>&1; tail -c 250 $O/msm_soak_g2_2p18_60_final.json; echo
