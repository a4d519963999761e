#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of one csrc/*.hip file, from hipcc -Rpass-analysis=kernel-resource-usage.
usage: tools/kernel_resources.py vq_assign_filter.hip [substring ...]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynamicvectorquantization_amd", "csrc")


def main():
    src = sys.argv[1]
    pats = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-ffp-contract=off",
           "-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + \
          [a for a in os.environ.get("EXTRA", "").split() if a]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(\S[^:]*): (\S+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = name.split("(")[0]
        if pats and not any(p in name for p in pats):
            continue
        print("%-70s vgpr %4s agpr %4s spill %4s scratch %6s lds %7s occ %s" % (
            name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
            r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
