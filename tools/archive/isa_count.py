#!/usr/bin/env python3
"""Instruction-issue budget of a kernel from a hipcc -save-temps .s file: static instruction counts of the straight-line
regions before / inside / after the innermost back-edge region that holds the MFMAs (prologue / code loop / epilogue of the
pass-1 kernels), split by issue class.  With the loop trip count this gives the instructions one wave issues per block --
the number DESIGN.md section 5.1 prices against the one-instruction-per-SIMD-per-4-cycles issue rate.
  python tools/isa_count.py file.s <mangled-name-substring> [trip_count]"""
import re
import sys


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("global_load_lds", "buffer_load") ) and "lds" in op: return "lds_dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")): return "wait/nop/barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    trips = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    s = open(path).read()
    names = [l.split(":")[0] for l in s.splitlines() if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
    for name in names:
        i = s.index("\n" + name + ":")
        j = s.index(".Lfunc_end", i)
        lines = []
        for l in s[i:j].splitlines()[2:]:
            t = l.split("//")[0].split(";")[0].strip()
            if not t or t.startswith((".", "#")) and not t.endswith(":"):
                continue
            lines.append(t)
        labels = {t[:-1]: k for k, t in enumerate(lines) if t.endswith(":")}
        best = None
        for k, t in enumerate(lines):
            m = re.match(r"s_cbranch_\w+ (\S+)", t) or re.match(r"s_branch (\S+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                a = labels[m.group(1)]
                nm = sum(1 for x in lines[a:k] if x.startswith("v_mfma"))
                if nm and (best is None or (k - a) < (best[1] - best[0])):
                    best = (a, k, nm)
        def count(seg):
            c = {}
            for t in seg:
                if t.endswith(":"): continue
                c[klass(t.split()[0])] = c.get(klass(t.split()[0]), 0) + 1
            c["issued"] = sum(v for kk, v in c.items() if kk != "wait/nop/barrier")
            return c
        if best is None:
            print(name, "no MFMA loop; whole kernel:", count(lines)); continue
        a, k, nm = best
        pro, loop, epi = count(lines[:a]), count(lines[a:k + 1]), count(lines[k + 1:])
        print(name[:60])
        print("  before loop (static):", pro)
        print("  loop body           :", loop)
        print("  after loop (static) :", epi)
        per_block = pro["issued"] + trips * loop["issued"] + epi["issued"]
        print("  issued per block at %d trips (upper bound: branches not taken count too): %d -> %d cycles at 4 cycles/instruction/SIMD"
              % (trips, per_block, 4 * per_block))


if __name__ == "__main__":
    main()
