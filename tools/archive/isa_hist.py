#!/usr/bin/env python3
"""Opcode histogram of the regions before / after the MFMA code loop of a pass-1 kernel in a hipcc -save-temps .s file.
  python tools/isa_hist.py file.s <mangled-name-prefix>"""
import collections
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
name = [l.split(":")[0] for l in s.splitlines() if l.startswith(pat) and ":" in l][0]
i = s.index("\n" + name + ":")
j = s.index(".Lfunc_end", i)
lines = [l.split("//")[0].split(";")[0].strip() for l in s[i:j].splitlines()[2:]]
lines = [l for l in lines if l and not (l.startswith((".", "#")) and not l.endswith(":"))]
mf = [k for k, l in enumerate(lines) if l.startswith("v_mfma_f32_16x16x32")]
k = mf[-1]
while not lines[k].startswith("s_cbranch"):
    k += 1
post = lines[k + 1:]
a = mf[-32]
while not lines[a].endswith(":"):
    a -= 1
pre = lines[:a]
for nm, seg in (("before the loop", pre), ("after the loop", post)):
    c = collections.Counter(l.split()[0] for l in seg if not l.endswith(":"))
    print(nm, sum(c.values()))
    print("  " + ", ".join("%d %s" % (n, op) for op, n in c.most_common(48)))
