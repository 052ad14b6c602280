"""Prints the kernel timeline of ONE proof from a rocprofv3 --kernel-trace CSV (dev aid).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/bench_proof.py --proofs 6
    python tools/proof_timeline.py DIR [which_proof_from_end]
A proof starts at a k_spmv launch; kernels are listed with start / end relative to the first kernel of the proof."""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


def main():
    d = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?")),
                  r.get("Workgroup_Size", ""), r.get("Grid_Size", "")) for r in rows))
    spmv = [i for i, e in enumerate(ev) if e[2].startswith("k_spmv")]
    i0 = spmv[-which]
    i1 = spmv[-which + 1] if which > 1 else len(ev)
    # the proof's first kernels (digits / sort of A) may precede k_spmv: start from the previous proof's last kernel end
    t0 = ev[i0][0]
    lo = i0
    while lo > 0 and ev[lo - 1][0] > t0 - 300000 and not ev[lo - 1][2].startswith("k_wsum_bits"):
        lo -= 1
    t0 = ev[lo][0]
    print("file", f)
    for s, e, n, q, wg, g in ev[lo:i1]:
        print("%9.1f %9.1f %8.1f us  q%-3s %-44s grid %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n, g))


if __name__ == "__main__":
    main()
