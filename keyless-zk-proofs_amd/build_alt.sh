#!/bin/bash
# A/B builds: keyless-zk-proofs_amd/build_alt.sh NAME "-DFLAG ..."  ->  alt/libk16_NAME.so (msm_g1 / msm_g2 recompiled with
# the flags, everything else from the current objects); load with K16_LIB_PATH
set -e
cd "$(dirname "$0")"
name=$1; flags=$2
mkdir -p alt
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result $flags"
/opt/rocm/bin/hipcc $F -c csrc/msm_g1.hip -o alt/msm_g1_$name.o &
/opt/rocm/bin/hipcc $F -c csrc/msm_g2.hip -o alt/msm_g2_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt/libk16_$name.so csrc/ctx.o csrc/msm_api.o csrc/msm_classes.o alt/msm_g1_$name.o alt/msm_g2_$name.o csrc/msm_sharded.o csrc/ntt.o csrc/prover.o csrc/verify.o csrc/fullprover.o -ldl
ls -la alt/libk16_$name.so
