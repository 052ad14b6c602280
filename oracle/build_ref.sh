#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.
# Compiles the REFERENCE's own Fq / Fr / Fq2 field implementation (generic
# GMP-mpn backend) from the sources where they lie under /root/reference into
# oracle/_ref/libref_field.so.  No reference build system is run, no stand-in
# headers are written: only files whose includes resolve in this image are
# compiled (gmp.h comes from /opt/conda/include, libgmp from the system).
# The reference's curve/multiexp/fft/groth16 sources include oneTBB,
# scope_guard.hpp and nlohmann/json.hpp, none of which exist in this image, so
# those are UNBUILDABLE here and are pinned through golden vectors instead
# ("Oracle").
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
RS="${REFERENCE_ROOT:-/root/reference}/rust-rapidsnark/rapidsnark/src"
if [ ! -d "$RS" ]; then
    echo "build_ref.sh: $RS not present; skipping (prebuilt oracle/_ref is used if it exists)"
    exit 0
fi
mkdir -p "$HERE/_ref"
GMP_SO=/usr/lib/x86_64-linux-gnu/libgmp.so.10
g++ -std=c++17 -O2 -fPIC -shared -w \
    -I "$RS" -I "$RS/.." -idirafter /opt/conda/include \
    "$HERE/ref_field_harness.cpp" \
    "$RS/fq.cpp" "$RS/fr.cpp" "$RS/fq_raw_generic.cpp" "$RS/fr_raw_generic.cpp" \
    "$RS/fq_generic.cpp" "$RS/fr_generic.cpp" "$RS/f2field.cpp" "$RS/splitparstr.cpp" \
    "$GMP_SO" -o "$HERE/_ref/libref_field.so"
echo "built $HERE/_ref/libref_field.so"
