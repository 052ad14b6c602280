#!/bin/bash
# builds /tmp/sortlab (or $1) against the in-tree libk16.so; extra hipcc flags via LABFLAGS
out=${1:-/tmp/sortlab}
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $LABFLAGS -I keyless-zk-proofs_amd/csrc tools/lab/sortlab.hip \
    -L keyless-zk-proofs_amd -lk16 -Wl,-rpath,$PWD/keyless-zk-proofs_amd -o "$out" 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|generated"; exit 0
