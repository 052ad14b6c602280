"""The verifier's pairing code (keyless-zk-proofs_amd/csrc/bn254_pairing.h) is __host__ __device__: this runs the exact
source the kernels compile on the CPU and compares Miller-loop values and pairings with the CPU oracle (no GPU needed)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import pymodel as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_pairing_header_matches_oracle_on_the_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "pairing_check")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                           "-I", os.path.join(ROOT, "keyless-zk-proofs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "pairing_check.cpp"), "-o", exe], timeout=900)
    g1 = ol.gen_points(0, 10, 6)
    g2 = ol.gen_points(1, 20, 6)
    g1[4] = 0                       # e(0, Q) = 1
    g2[5] = 0                       # e(P, 0) = 1
    pairs = np.concatenate([g1, g2], axis=1)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    pairs.tofile(fin)
    out = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK 6"), out.stdout + out.stderr
    got = np.fromfile(fout, dtype=np.uint8).reshape(6, 2, 384)
    for i in range(6):
        assert got[i, 0].tobytes() == ol.miller(bytes(g1[i]), bytes(g2[i])), i
        assert got[i, 1].tobytes() == ol.pairing(bytes(g1[i]), bytes(g2[i])), i
