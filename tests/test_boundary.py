"""Drop-in boundary: the C-ABI library loads and exports every symbol include/k16.h declares, the
C++ FullProver facade keeps the reference's ABI, and both fail loudly (never fall back) without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "keyless-zk-proofs_amd")
LIB = os.path.join(PKG, "libk16.so")
HARNESS_SRC = os.path.join(ROOT, "tests", "cpp", "fullprover_harness.cpp")


def _has_gpu():
    try:
        import k16
        c = k16.Context(0)
        c.close()
        return True
    except Exception:
        return False


def build_harness(tmp_path, testing=False):
    """The C++ program standing for the Rust crate.  testing=True links the TESTING build of the library
    (libk16_testing.so, the only one with the K16_FAULT_INJECT hooks)."""
    exe = str(tmp_path / ("fullprover_harness_testing" if testing else "fullprover_harness"))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), HARNESS_SRC,
                           "-L", PKG, "-lk16_testing" if testing else "-lk16", "-Wl,-rpath," + PKG, "-pthread", "-o", exe])
    return exe


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    hdr = open(os.path.join(ROOT, "include", "k16.h")).read()
    declared = set(re.findall(r"\b(k16_[a-z0-9_]+)\s*\(", hdr))
    import k16
    assert declared == set(k16.SYMBOLS), declared ^ set(k16.SYMBOLS)
    L = ctypes.CDLL(LIB)
    for s in sorted(declared):
        assert hasattr(L, s), s
    # C++ facade: Itanium-mangled entry points bindgen binds (rust-rapidsnark/build.rs:142-146)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", LIB]).decode()
    for m in ("_ZN10FullProverC1EPKc", "_ZN10FullProverD1Ev", "_ZNK10FullProver5proveEPKc",
              "_ZN14ProverResponseC1E11ProverError", "_ZN14ProverResponseD1Ev"):
        assert m in syms, m


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU behaviour")
def test_no_gpu_fails_loudly_no_fallback(tmp_path, toy_paths):
    import k16
    with pytest.raises(k16.K16Error) as ei:
        k16.Context(0)
    assert ei.value.rc == -1
    zkey, wtns, _ = toy_paths
    exe = build_harness(tmp_path)
    out = subprocess.run([exe, zkey, wtns], capture_output=True, text=True, timeout=120)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=1"                       # not OK
    assert lines[1].startswith("type=1 error=1")       # ERROR / PROVER_NOT_READY (fullprover.cpp:117-121)
    assert lines[2] == ""                              # raw_json = ""
    assert "no CPU fallback" in out.stderr


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under keyless-zk-proofs_amd/ or include/ may reference it."""
    bad = []
    for base in (PKG, os.path.join(ROOT, "include")):
        for dp, _, fns in os.walk(base):
            for fn in fns:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"oracle_lib|liboracle|oracle/|bn254_ref|ora_[a-z]+\(", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


@pytest.mark.gpu
def test_fullprover_facade_on_gpu(tmp_path, toy_paths):
    import json
    import bn254_pairing as bp
    zkey, wtns, vk = toy_paths
    exe = build_harness(tmp_path)
    out = subprocess.run([exe, zkey, wtns, "3"], capture_output=True, text=True, timeout=300)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr
    proofs = []
    for k in range(3):
        assert lines[1 + 2 * k].startswith("type=0 error=0 ms="), lines
        proofs.append(lines[2 + 2 * k])
    assert len(set(proofs)) == 3                      # fresh CSPRNG blinding each time
    for js in proofs:
        assert json.loads(js)["protocol"] == "groth16"
        assert bp.verify_json(vk, js, [2])            # reference criterion: prover_handler.rs:279-290
    # error mapping (fullprover.cpp:91-100, 117-121, 216-221)
    missing = subprocess.run([exe, str(tmp_path / "nope.zkey"), wtns], capture_output=True, text=True, timeout=120)
    assert missing.stdout.splitlines()[0] == "state=1"          # ZKEY_FILE_LOAD_ERROR
    assert missing.stdout.splitlines()[1].startswith("type=1 error=1")
    wrong = subprocess.run([exe, wtns, wtns], capture_output=True, text=True, timeout=120)
    assert wrong.stdout.splitlines()[0] == "state=2"            # UNSUPPORTED_ZKEY_CURVE (wrong container type)
    garbage = tmp_path / "bad.wtns"
    garbage.write_bytes(b"wtns" + b"\x02\0\0\0" + b"\x01\0\0\0" + b"\xff" * 20)
    bad = subprocess.run([exe, zkey, str(garbage)], capture_output=True, text=True, timeout=120)
    assert bad.stdout.splitlines()[1].startswith("type=1 error=2")  # INVALID_INPUT
    # a witness over another prime -> WITNESS_GENERATION_INVALID_CURVE
    w = bytearray(open(wtns, "rb").read())
    i = w.index(bytes.fromhex("010000f093f5e143"))
    w[i] ^= 0x02
    other = tmp_path / "othercurve.wtns"
    other.write_bytes(bytes(w))
    oc = subprocess.run([exe, zkey, str(other)], capture_output=True, text=True, timeout=120)
    assert oc.stdout.splitlines()[1].startswith("type=1 error=3")


@pytest.mark.gpu
def test_fullprover_pool_concurrent_callers(tmp_path, toy_paths):
    """K16_DEVICES puts several provers behind ONE FullProver; concurrent prove() calls are safe and
    every proof verifies.  Here three provers share GPU 0 (a multi-GPU node lists 0,1,...,7)."""
    import json
    import bn254_pairing as bp
    zkey, wtns, vk = toy_paths
    exe = build_harness(tmp_path)
    env = dict(os.environ, K16_DEVICES="0,0,0")
    out = subprocess.run([exe, zkey, wtns, "2", "4"], capture_output=True, text=True, timeout=300, env=env)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr
    proofs = [lines[2 + 2 * k] for k in range(8)]      # 4 threads x 2 proofs
    for k in range(8):
        assert lines[1 + 2 * k].startswith("type=0 error=0 ms="), lines
    assert len(set(proofs)) == 8
    for js in proofs[:4]:
        assert json.loads(js)["protocol"] == "groth16"
        assert bp.verify_json(vk, js, [2])
    # a pool that names a device that does not exist fails in the constructor, loudly
    bad = subprocess.run([exe, zkey, wtns], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, K16_DEVICES="0,99"))
    assert bad.stdout.splitlines()[0] == "state=1"
    assert "no usable" in bad.stderr


@pytest.mark.gpu
def test_fullprover_pool_with_the_witness_in_memory(tmp_path, toy_paths):
    """k16_fullprover_prove_mem: the facade's pool (K16_DEVICES=0,0,0), four caller threads, the witness handed over in
    memory instead of as a file path -- what a multi-GPU service binds.  Every proof verifies
    (prover_handler.rs:279-290); a prover that failed to construct answers PROVER_NOT_READY; a witness shorter than the
    circuit is the caller's fault."""
    import json
    import bn254_pairing as bp
    zkey, wtns, vk = toy_paths
    exe = build_harness(tmp_path)
    env = dict(os.environ, K16_DEVICES="0,0,0", K16_HARNESS_MEM="1")
    out = subprocess.run([exe, zkey, wtns, "2", "4"], capture_output=True, text=True, timeout=300, env=env)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr
    proofs = [lines[2 + 2 * k] for k in range(8)]
    for k in range(8):
        assert lines[1 + 2 * k].startswith("type=0 error=0 ms="), lines
    assert len(set(proofs)) == 8
    for js in proofs[:4]:
        assert json.loads(js)["protocol"] == "groth16"
        assert bp.verify_json(vk, js, [2])
    missing = subprocess.run([exe, str(tmp_path / "nope.zkey"), wtns], capture_output=True, text=True, timeout=120, env=env)
    assert missing.stdout.splitlines()[0] == "state=1"
    assert missing.stdout.splitlines()[1].startswith("type=1 error=1")      # PROVER_NOT_READY
    w = open(wtns, "rb").read()
    short = tmp_path / "short.wtns"                                            # section 2 cut to two values
    i = w.index(b"\x02\x00\x00\x00", 12 + 12 + 40)
    short.write_bytes(w[:i] + b"\x02\x00\x00\x00" + (64).to_bytes(8, "little") + w[i + 12:i + 12 + 64])
    bad = subprocess.run([exe, zkey, str(short)], capture_output=True, text=True, timeout=120, env=env)
    assert bad.stdout.splitlines()[1].startswith("type=1 error=2"), bad.stdout   # INVALID_INPUT


@pytest.mark.gpu
def test_fullprover_device_fault_is_not_the_callers_fault(tmp_path, toy_paths):
    """A HIP failure inside prove() (injected: K16_FAULT_INJECT) is reported as PROVER_NOT_READY -- the class the service
    retries -- not INVALID_INPUT, the slot is rebuilt in a fresh context before it is handed out again, and the proofs
    after it are good (RS/fullprover.cpp:91-125 maps only its own exception classes; a GPU build adds this one)."""
    import json
    import bn254_pairing as bp
    zkey, wtns, vk = toy_paths
    exe = build_harness(tmp_path, testing=True)
    env = dict(os.environ, K16_FAULT_INJECT="hip_after_msm:2")
    out = subprocess.run([exe, zkey, wtns, "4"], capture_output=True, text=True, timeout=300, env=env)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr
    assert lines[1].startswith("type=0 error=0")
    assert lines[3].startswith("type=1 error=1"), lines           # ERROR / PROVER_NOT_READY
    assert lines[4] == ""
    for k in (5, 7):
        assert lines[k].startswith("type=0 error=0"), lines
        assert json.loads(lines[k + 1])["protocol"] == "groth16"
        assert bp.verify_json(vk, lines[k + 1], [2])


@pytest.mark.gpu
def test_fullprover_exception_in_prove_releases_the_slot(tmp_path, toy_paths):
    """An exception inside a proof (injected std::bad_alloc, testing build) ends at the C ABI's firewall as
    K16_ERR_NOMEM -> PROVER_NOT_READY, and the ONE slot of the default pool is given back: the next prove() succeeds
    instead of waiting for ever in acquire() (round-2 advisor finding).  The production build ignores the variable."""
    import json
    import bn254_pairing as bp
    zkey, wtns, vk = toy_paths
    exe = build_harness(tmp_path, testing=True)
    env = dict(os.environ, K16_FAULT_INJECT="bad_alloc_in_prove:2")
    env.pop("K16_DEVICES", None)
    out = subprocess.run([exe, zkey, wtns, "4"], capture_output=True, text=True, timeout=120, env=env)
    lines = out.stdout.splitlines()
    assert lines[0] == "state=0", out.stderr
    assert lines[1].startswith("type=0 error=0")
    assert lines[3].startswith("type=1 error=1"), lines           # ERROR / PROVER_NOT_READY
    assert lines[4] == ""
    for k in (5, 7):                                              # the pool's only slot is alive and proves correctly
        assert lines[k].startswith("type=0 error=0"), lines
        assert bp.verify_json(vk, lines[k + 1], [2])
    prod = build_harness(tmp_path)
    out = subprocess.run([prod, zkey, wtns, "3"], capture_output=True, text=True, timeout=120,
                         env=dict(env, K16_FAULT_INJECT="hip_after_msm"))
    lines = out.stdout.splitlines()
    assert [lines[1 + 2 * k].startswith("type=0 error=0") for k in range(3)] == [True] * 3, lines


@pytest.mark.gpu
def test_fullprover_pool_over_all_devices(tmp_path, toy_paths):
    """K16_DEVICES=all builds the pool from the device count (one resident key per GPU); with two or more GPUs the
    concurrent callers really prove on different ordinals (K16_LOG names the device of every proof)."""
    import bn254_pairing as bp
    import k16
    zkey, wtns, vk = toy_paths
    exe = build_harness(tmp_path)
    n_dev = k16.load().k16_device_count()
    assert n_dev >= 1
    env = dict(os.environ, K16_DEVICES="all", K16_LOG="1")
    out = subprocess.run([exe, zkey, wtns, "3", str(2 * n_dev)], capture_output=True, text=True, timeout=300, env=env)
    lines = [l for l in out.stdout.splitlines() if not l.startswith('{"level"')]
    assert lines[0] == "state=0", out.stderr
    n = 3 * 2 * n_dev
    for k in range(n):
        assert lines[1 + 2 * k].startswith("type=0 error=0"), lines
    assert bp.verify_json(vk, lines[2], [2])
    used = set(re.findall(r"on device (\d+)", out.stdout))
    assert used == {str(d) for d in range(n_dev)}, (used, n_dev)
    if n_dev < 2:
        pytest.skip("one GPU here: the two-ordinal leg needs a multi-GPU node (the pool over all = [0] was exercised)")
