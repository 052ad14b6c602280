"""Reduces rocprofv3 --pmc counter CSVs (one counter per pass, as MI355X_MICROARCH.md prescribes) to the per-launch HBM-side
traffic of the MSM's dominant kernel and writes profiles/pmc_traffic.json, which bench.py quotes as `roofline.traffic`.

    python tools/pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [out.json]
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "profiles", "pmc_traffic.json")
    fetch, write = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    acc = [k for k in fetch if "k_accumulate" in k and "Eng9" in k]
    if not acc:
        sys.exit("no k_accumulate<Eng9> rows in %s" % fd)
    k = acc[0]
    f_kb, n = fetch[k]
    w_kb = write.get(k, (0.0, 0))[0]
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (one counter per pass), "
                  "K16_BENCH_DEPTH=1 bench.py --steps 3 --warmup 1 --proofs 0; per-kernel averages in profiles/r03/pmc_*_per_kernel.txt (tools/pmc_kernel.py on the same passes)",
        "kernel": k[:60],
        "launches_averaged": n,
        "FETCH_SIZE_KB_raw": f_kb,
        "WRITE_SIZE_KB": w_kb,
        "correction": "FETCH_SIZE doubled: on gfx950 it tallies 128-B requests at 64 B for wide coalesced streaming reads "
                      "(MI355X_MICROARCH.md, HBM section).  ASSUMPTION: the accumulation's reads are 64-B row GATHERS (4 x 16 B "
                      "per lane), not a streaming read; whether they are tallied the same way was not verified, so the true "
                      "figure lies between FETCH_SIZE x 1 and x 2.  WRITE_SIZE is exact for 16-B/lane stores.",
        "msm_accumulate_hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
        "msm_accumulate_hbm_bytes_per_launch_uncorrected": int(f_kb * 1024 + w_kb * 1024),
        "note": "memory-side (fabric) requests: Infinity-Cache hits are counted.  The 64 MB point table is gathered once "
                "per non-zero digit (16 x per point) and is MALL-resident, so this is mostly MALL traffic, not HBM re-reads.",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
