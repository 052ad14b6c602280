#!/bin/bash
# builds tools/lab/affinelab (or $1); header-only use of the product's field / curve code, no library needed
out=${1:-tools/lab/affinelab}
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $LABFLAGS -I include -I keyless-zk-proofs_amd/csrc tools/lab/affinelab.hip \
    -o "$out" 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|generated"; exit 0
