"""How two provers sharing one GPU interleave (dev aid): from a rocprofv3 --kernel-trace CSV of a throughput-mode run, the H MSM's
bucket accumulations (the longest k_accumulate<Eng9> launches) of both provers, how much of each overlaps another one, and the
share of the run in which no accumulation kernel of either prover is resident (the multiply-add units then idle).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/bench_proof.py --proofs 60 --concurrent 2 --no-stats
    python3 tools/two_prover_phases.py DIR [proofs_from_the_end]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    f = max(files, key=os.path.getsize)
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0],
                 r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
    spmv = [e for e in ev if e[2].startswith("k_spmv")]
    if len(spmv) < 8:
        sys.exit("no proofs in the trace")
    # the stretch: the last `last` proofs of the trace (the throughput leg runs last)
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    last = min(last, len(spmv) - 1)
    t_lo, t_hi = spmv[-last - 1][0], spmv[-1][0]
    n_proofs = last
    acc = [e for e in ev if e[2].startswith("k_accumulate") and e[1] > t_lo and e[0] < t_hi]
    hacc = [e for e in acc if e[2].startswith("k_accumulate<Eng9>") and e[1] - e[0] > 1200000]
    print("file %s\nsteady stretch %.1f ms, %d proofs: %.2f ms per proof = %.1f proofs/s" % (f, (t_hi - t_lo) / 1e6, n_proofs, (t_hi - t_lo) / 1e6 / n_proofs, n_proofs / ((t_hi - t_lo) / 1e9)))
    print("H accumulations: %d, mean %.2f ms (min %.2f, max %.2f)" % (len(hacc), sum(e[1] - e[0] for e in hacc) / 1e6 / max(1, len(hacc)),
                                                                       min(e[1] - e[0] for e in hacc) / 1e6, max(e[1] - e[0] for e in hacc) / 1e6))
    ov = 0
    for i, a in enumerate(hacc):
        for b in hacc[i + 1:]:
            ov += max(0, min(a[1], b[1]) - max(a[0], b[0]))
    print("time in which TWO H accumulations overlap: %.1f %% of the stretch" % (100.0 * ov / (t_hi - t_lo)))
    # union of all accumulation kernels
    iv = sorted((max(e[0], t_lo), min(e[1], t_hi)) for e in acc)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    print("no accumulation kernel resident: %.1f %% of the stretch (%.2f ms per proof)" % (100.0 * (1 - busy / (t_hi - t_lo)), (t_hi - t_lo - busy) / 1e6 / n_proofs))
    hu = sorted((e[0], e[1]) for e in hacc)
    hb = 0
    cur_s = cur_e = None
    for s, e in hu:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                hb += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        hb += cur_e - cur_s
    print("an H accumulation resident: %.1f %% of the stretch" % (100.0 * hb / (t_hi - t_lo)))
    print("\nH accumulations (ms from the stretch's start; stream):")
    for s, e, n, q in hacc[:16]:
        print("  %8.2f -> %8.2f  (%.2f ms)  stream %s" % ((s - t_lo) / 1e6, (e - t_lo) / 1e6, (e - s) / 1e6, q))


if __name__ == "__main__":
    main()


def window(d, lo_ms, hi_ms, last=80, min_us=40):
    """kernels longer than min_us in [lo_ms, hi_ms] of the stretch: python3 -c 'import two_prover_phases as t; t.window(DIR, 11, 17)'"""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    f = max(files, key=os.path.getsize)
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0],
                 r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("Grid_Size", "")) for r in rows)
    spmv = [e for e in ev if e[2].startswith("k_spmv")]
    t_lo = spmv[-min(last, len(spmv) - 1) - 1][0]
    for s, e, n, q, g in ev:
        if e - s >= min_us * 1000 and (e - t_lo) / 1e6 >= lo_ms and (s - t_lo) / 1e6 <= hi_ms:
            print("%8.2f -> %8.2f  %7.0f us  s%-3s %-44s %s" % ((s - t_lo) / 1e6, (e - t_lo) / 1e6, (e - s) / 1e3, q, n[:44], g))
