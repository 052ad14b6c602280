#!/bin/bash
# A/B builds: keyless-zk-proofs_amd/build_alt.sh NAME "-DFLAG ..." [unit ...]  ->  alt/libk16_NAME.so: the named units
# (default: msm_g1 msm_g2; any of ctx msm_api msm_classes msm_g1 msm_g2 msm_sharded ntt prover verify) recompiled with the
# flags, everything else from the current objects; load with K16_LIB_PATH
set -e
cd "$(dirname "$0")"
name=$1; flags=$2; shift; shift
units=${*:-msm_g1 msm_g2}
mkdir -p alt
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result $flags"
objs=""
for u in ctx msm_api msm_classes msm_g1 msm_g2 msm_sharded ntt prover verify; do
  if [[ " $units " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc $F -c csrc/$u.hip -o alt/${u}_$name.o &
    objs="$objs alt/${u}_$name.o"
  else
    objs="$objs csrc/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt/libk16_$name.so $objs csrc/fullprover.o -ldl
ls -la alt/libk16_$name.so
