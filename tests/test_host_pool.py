"""CPU test of the process-wide host thread pool (keyless-zk-proofs_amd/csrc/host_pool.h): the header is
compiled with g++ into tests/cpp/host_pool_check.cpp -- several callers running jobs at once, every task exactly once,
run() returning only when its own job is done -- once plainly and once under ThreadSanitizer (sanitizers run on the CPU
build only)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tsan", [False, True])
def test_host_pool_runs_concurrent_jobs(tmp_path, tsan):
    exe = str(tmp_path / ("host_pool_check" + ("_tsan" if tsan else "")))
    flags = ["-fsanitize=thread", "-O1", "-g"] if tsan else ["-O2"]
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-pthread"] + flags +
                          ["-I", os.path.join(ROOT, "keyless-zk-proofs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "host_pool_check.cpp"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
