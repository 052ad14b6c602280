"""Feeds tools/lab/ubench_shoup its constants and checks both product chains against Python integers (dev aid)."""
import subprocess, sys
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
def limbs(v): return [(v >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
def val(l): return sum(x << (29 * i) for i, x in enumerate(l))
w = pow(5, (R - 1) >> 20, R)
wq = (w << 261) // R
xs = [pow(7, 1000 + i, R) for i in range(4)]
xs[0] ^= 0  # lane 0 of block 0: threadIdx.x & 1 == 0
inp = limbs(w) + limbs(wq) + sum((limbs(x) for x in xs), [])
out = subprocess.run([sys.argv[1]], input=" ".join(map(str, inp)), capture_output=True, text=True).stdout
print(out)
iters = 512
lines = [l for l in out.splitlines() if l.startswith("limbs")]
m = [int(t) for t in lines[0].split(":")[1].split()]
s = [int(t) for t in lines[1].split(":")[1].split()]
Rp = pow(2, 261, R)
for v in range(4):
    # montgomery: x <- w x / 2^261 each step, with w taken as is
    want_m = xs[v] * pow(w * pow(Rp, -1, R), iters, R) % R
    want_s = xs[v] * pow(w, iters, R) % R
    gm, gs = val(m[9 * v:9 * v + 9]), val(s[9 * v:9 * v + 9])
    print("value %d: montgomery %s (%.2f r)   shoup %s (%.2f r)" % (v, gm % R == want_m, gm / R, gs % R == want_s, gs / R))
