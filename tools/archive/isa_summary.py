#!/usr/bin/env python3
"""Summary of a kernel's instruction stream from a hipcc -save-temps .s file: the run-length-compressed sequence of
memory / wait / barrier / MFMA instructions (optionally only up to the first MFMA).  Used to check by eye that
hand-placed waits and DMA issues sit where the source puts them.
  python tools/isa_summary.py file.s <mangled-name-substring> [--prologue]"""
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    prologue = "--prologue" in sys.argv
    s = open(path).read()
    names = [l.split(":")[0] for l in s.splitlines() if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
    for name in names:
        i = s.index("\n" + name + ":")
        j = s.index(".Lfunc_end", i)
        body = s[i:j].splitlines()
        keep = ("global_load", "global_store", "global_atomic", "buffer_", "s_waitcnt", "s_barrier", "v_mfma", "ds_read",
                "ds_write", "s_sleep", "s_getreg", "s_setprio", "s_cbranch", "scratch_")
        seq = []
        for l in body:
            t = l.strip().split("//")[0].strip()
            if t.startswith(keep):
                op = t.split()[0]
                key = t if op in ("s_waitcnt", "s_setprio") else op
                seq.append(key)
        if prologue:
            k = next((n for n, t in enumerate(seq) if t.startswith("v_mfma")), len(seq))
            seq = seq[:k]
        comp = []
        for t in seq:
            if comp and comp[-1][0] == t:
                comp[-1][1] += 1
            else:
                comp.append([t, 1])
        print("==", name, len(body), "lines")
        print(" ".join("%s%s" % (k, ("*%d" % n) if n > 1 else "") for k, n in comp))


main()
