#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py keyless-zk-proofs_amd/csrc/msm_g1.hip [extra hipcc flags]"""
import re, subprocess, sys, os, tempfile

def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    out = tempfile.mktemp(suffix=".o")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", out] + extra,
                       capture_output=True, text=True)
    if os.path.exists(out):
        os.unlink(out)
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
            ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
    for b in blocks:
        name = b.split("\n")[0].strip()
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(.*", "", dn)[:72]
        vals = []
        for label, k in keys:
            m = re.search(k + r": (\d+)", b)
            vals.append("%s %5s" % (label, m.group(1) if m else "?"))
        print("%-72s %s" % (dn, "  ".join(vals)))

if __name__ == "__main__":
    main()
