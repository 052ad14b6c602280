"""Child process of tests/test_gpu_parity.py::test_prover_failure_leaves_no_stale_msms.

usage: fault_inject_child.py testing|production TMPDIR      (K16_LIB_PATH selects the library build)

testing     libk16_testing.so (-DK16_TESTING): an injected device fault / std::bad_alloc in the middle of a proof is
            reported as an error code (never an exception across the C ABI), leaves no MSM behind, and the next proof
            on the same prover is byte-equal to the oracle's.
production  libk16.so has no fault hooks: K16_FAULT_INJECT in the environment changes nothing.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "keyless-zk-proofs_amd"))

import k16            # noqa: E402
import oracle_lib as ol   # noqa: E402  (the checker)
import pymodel as pm  # noqa: E402
import zkey_builder as zb  # noqa: E402


def main():
    mode, tmp = sys.argv[1], sys.argv[2]
    assert os.path.basename(k16.LIB_PATH) == ("libk16_testing.so" if mode == "testing" else "libk16.so"), k16.LIB_PATH
    zk, wt = os.path.join(tmp, "s_%s.zkey" % mode), os.path.join(tmp, "s_%s.wtns" % mode)
    zb.build_zkey(zk, 3000, 2, 4096, 9000, seed=21)
    w = zb.build_wtns(wt, 3000, seed=22)
    r, s = pm.limbs(5), pm.limbs(6)
    want = ol.prove_files(zk, wt, r, s, nthreads=4)
    ctx = k16.Context(0)
    p = k16.Prover(ctx, zk)
    assert p.warmup_status() == 0
    assert p.prove_mem(w, r, s) == want
    for kind, rc_want, text in (("hip_after_msm", -2, "injected"), ("bad_alloc_in_prove", -8, "memory")):
        os.environ["K16_FAULT_INJECT"] = kind
        if mode == "production":
            assert p.prove_mem(w, r, s) == want        # the variable is not even read
        else:
            try:
                p.prove_mem(w, r, s)
                raise AssertionError("the injected fault did not surface")
            except k16.K16Error as e:
                assert e.rc == rc_want and text in str(e), (e.rc, str(e))
        del os.environ["K16_FAULT_INJECT"]
        assert ctx.msm_pending() == 0
        assert p.prove_mem(w, r, s) == want
        assert p.prove_file(wt, r, s) == want
    p.close()
    ctx.close()
    print("fault child OK (%s)" % mode)


if __name__ == "__main__":
    main()
