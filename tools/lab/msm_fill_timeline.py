"""Kernel timeline of the FIRST steps of the pipelined MSM bench's timed region (pipeline fill), from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --proofs 0 --no-cpu-baseline
    python3 tools/lab/msm_fill_timeline.py DIR [steps=20] [show_ms=6]
The timed region's accumulations are the last `steps` k_accumulate<Eng9> launches (bench.py's isolated pass of 3 MSMs follows: skipped)."""
import csv, glob, os, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
show = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(fs[-1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44],
             r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
acc = [e for e in ev if e[2].startswith("k_accumulate<Eng9>")]
# runs of accumulations separated by more than 2.5 ms of nothing: prewarm, warm-up, TIMED REGION (exactly `steps`), isolated pass, host leg
groups, cur = [], [acc[0]]
for a, b in zip(acc, acc[1:]):
    if b[0] - a[1] > 2500000:
        groups.append(cur)
        cur = []
    cur.append(b)
groups.append(cur)
print("runs of accumulations:", [len(g) for g in groups])
cand = [g for g in groups if len(g) in (steps, steps + 3)]   # (bench.py's isolated pass of 3 MSMs may follow within the 2.5 ms)
acc = cand[-1][:steps]
t_first = acc[0][0]
# the region starts with the memset / first sort kernel of the first MSM: walk back to the previous idle gap > 200 us
i0 = next(i for i, e in enumerate(ev) if e[0] == t_first)
lo = i0
while lo > 0 and ev[lo][0] - max(x[1] for x in ev[max(0, lo - 40):lo]) < 200000:
    lo -= 1
t0 = ev[lo][0]
print("timed region: first kernel at 0, first accumulation at %.1f us, last accumulation ends at %.1f us" % ((t_first - t0) / 1e3, (acc[-1][1] - t0) / 1e3))
for k, a in enumerate(acc):
    print("  accumulation %2d  %9.1f -> %9.1f  (%7.1f us)  q%s" % (k, (a[0] - t0) / 1e3, (a[1] - t0) / 1e3, (a[1] - a[0]) / 1e3, a[3]))
print("kernels of the first %.1f ms:" % show)
for s, e, n, q in ev[lo:]:
    if (s - t0) / 1e6 > show:
        break
    print("%9.1f %9.1f %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
