"""Which NUMA node is the GPU on, and what does pinning the host thread there do to the launch cost?"""
import glob, os, subprocess, sys, json
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    try:
        vendor = open(os.path.dirname(p) + "/vendor").read().strip()
        print(p, open(p).read().strip(), vendor)
    except Exception as e:
        print(p, e)
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip())
print("cpu now:", os.sched_getaffinity(0).__len__(), "allowed;")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for node in sorted(glob.glob("/sys/devices/system/node/node*")):
    cl = open(node + "/cpulist").read().strip()
    env = dict(os.environ, K16_BENCH_CPUS=cl)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(os.path.basename(node), cl[:30], "->", round(d["value"] / 1e6, 1), "M/s", d["ms_per_step"], d["host_ms"])
    except Exception as e:
        print(node, "failed", e, out.stderr[-300:])
