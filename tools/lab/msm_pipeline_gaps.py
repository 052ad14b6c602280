"""Gaps between consecutive bucket accumulations of the pipelined MSM bench, and what runs inside them (dev aid):
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --proofs 0 --no-cpu-baseline
    python3 tools/lab/msm_pipeline_gaps.py DIR"""
import csv, glob, os, sys
d = sys.argv[1]
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(fs[-1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]) for r in rows)
acc = [e for e in ev if e[2].startswith("k_accumulate<Eng9>")]
acc = acc[-25:]                      # the timed region's
tot = (acc[-1][1] - acc[0][0]) / 1e3
busy = sum(e[1] - e[0] for e in acc) / 1e3
print("last %d accumulations: span %.1f us, accumulation busy %.1f us (%.1f %%), mean duration %.1f us, mean period %.1f us" %
      (len(acc), tot, busy, 100 * busy / tot, busy / len(acc), tot / (len(acc) - 1)))
gaps = []
for a, b in zip(acc, acc[1:]):
    g = (b[0] - a[1]) / 1e3
    inside = sorted(set(e[2] for e in ev if e[0] < b[0] and e[1] > a[1] and not e[2].startswith("k_accumulate<Eng9>")))
    gaps.append(g)
    print("  gap %7.1f us   kernels overlapping it: %s" % (g, ", ".join(inside)[:200]))
print("mean gap %.1f us" % (sum(gaps) / len(gaps)))
# what overlaps the accumulations themselves
over = {}
for a in acc[5:15]:
    for e in ev:
        if e[0] < a[1] and e[1] > a[0] and not e[2].startswith("k_accumulate<Eng9>"):
            over[e[2]] = over.get(e[2], 0) + (min(e[1], a[1]) - max(e[0], a[0])) / 1e3
print("kernels running beside 10 accumulations (us of overlap, summed):")
for k, v in sorted(over.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-42s %8.1f" % (k, v))
