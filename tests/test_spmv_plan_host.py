"""CPU test of the SpMV layout planner (keyless-zk-proofs_amd/csrc/spmv_plan.h, host logic of k16_prover_create): the
header is compiled with g++ into tests/cpp/spmv_plan_check.cpp, which emulates k_spmv over the plan with small-modulus
arithmetic and compares every output row with the reference's walk over the coefficient list (RS/groth16.cpp:137-156) --
empty matrices, rows of exactly 63 / 64 / 65 entries, rows of thousands, every row long, an index out of range."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spmv_plan_matches_the_coefficient_walk(tmp_path):
    exe = str(tmp_path / "spmv_plan_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "keyless-zk-proofs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "spmv_plan_check.cpp"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
