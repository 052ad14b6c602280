"""Host cross-check of the radix-2^29 field and G1 formulas (bn254_fq9.h) against the canonical
8x32-bit implementation: both are __host__ __device__, so the exact device arithmetic is exercised
on the CPU -- values, lazy-reduction bounds and every exceptional case of the XYZZ formulas."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_fq9_matches_canonical_field_and_curve(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "fq9_check")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                           "-I", os.path.join(ROOT, "keyless-zk-proofs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "fq9_check.cpp"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
