#!/bin/bash
# Round 6 fuzz / soak on the final build (provers of the soak share ONE resident key; NTT with the stage-major twiddle tables)
O=gpurun_out/r06; mkdir -p $O
python3 tools/prove_soak.py ${1:-4000} 2 > $O/prove_soak_keyless_shape_shared_key.log 2>&1; tail -1 $O/prove_soak_keyless_shape_shared_key.log
python3 tools/msm_fuzz.py 1500 61 > $O/msm_fuzz_1500_seed61.json 2>&1; tail -c 250 $O/msm_fuzz_1500_seed61.json; echo
python3 tools/ntt_fuzz.py 600 61 > $O/ntt_fuzz_600_seed61.json 2>&1; tail -c 250 $O/ntt_fuzz_600_seed61.json; echo
python3 tools/prove_fuzz.py 400 61 > $O/prove_fuzz_400_seed61.json 2>&1; tail -c 250 $O/prove_fuzz_400_seed61.json; echo
python3 tools/classes_fuzz.py 150 61 > $O/classes_fuzz_150_seed61.json 2>&1; tail -c 250 $O/classes_fuzz_150_seed61.json; echo
python3 tools/ntt_soak.py 21 100 > $O/ntt_soak_2p21_100.json 2>&1; tail -c 250 $O/ntt_soak_2p21_100.json; echo
