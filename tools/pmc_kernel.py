"""Averages rocprofv3 --pmc counter values per kernel:  python tools/pmc_kernel.py <dir> [substring]"""
import csv
import glob
import os
import sys

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:70], r["Counter_Name"])
        if sub and sub not in r["Kernel_Name"]:
            continue
        t = acc.setdefault(k, [0.0, 0])
        t[0] += float(r["Counter_Value"])
        t[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-72s %-22s avg %14.2f  (n=%d)" % (k, c, v / n, n))
