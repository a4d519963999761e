#!/usr/bin/env python3
"""ISA check for the hand-issued asynchronous loads of the conv prologue (vq_assign_filter_kernel<256, SEL, true> and the small-batch
vq_assign_filter_split_conv_kernel<256, SEL>): the x values of
k-step g are loaded with `asm volatile("global_load_dword ... nt")` three k-steps before they are used and waited for with a COUNTED
s_waitcnt; hipcc believes the destination registers are defined at the asm statement, so nothing it schedules between the load and
its covering wait may read, copy, spill or overwrite them.  The 8 loads of group g are covered by the (g + 1)-th vmcnt wait after
the first load (compiler-inserted waits in between only make the counted ones stricter).  Run by __graft_entry__.build().
  python tools/isa_hazard_check.py [file.s]      (without an argument: compiles csrc/vq_assign_filter.hip to gfx950 assembly first)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assemble(out_path):
    src = os.path.join(ROOT, "dynamicvectorquantization_amd", "csrc", "vq_assign_filter.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-S", "-o", out_path, src], stderr=subprocess.DEVNULL)


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    s = open(path).read()
    names = [l.split(":")[0] for l in s.splitlines()
             if l.startswith("_Z23vq_assign_filter_kernelILi256") and "ELb1ELb0EE" in l.split(":")[0]]   # <256, SEL, CONV = true, FOLD = false>
    assert names, "no CONV instantiation of vq_assign_filter_kernel in %s" % path
    split = sorted({l.split(":")[0] for l in s.splitlines()               # the small-batch form with the same prologue
                    if l.startswith("_Z34vq_assign_filter_split_conv_kernelILi256") and ":" in l})
    assert len(split) == 2, "expected the SEL = 0 / 1 instantiations of vq_assign_filter_split_conv_kernel, found %r" % split
    names += split
    report = []
    for name in names:
        i = s.index("\n" + name + ":")
        j = s.index(".Lfunc_end", i)
        body = [l.split("//")[0].split(";")[0].strip() for l in s[i:j].splitlines()]
        body = [l for l in body if l and not l.startswith(".")]
        loads = [k for k, l in enumerate(body) if re.match(r"global_load_dword \S+ \S+ off nt", l.replace(",", ""))]
        assert len(loads) == 128, (name, len(loads))
        waits = [k for k, l in enumerate(body) if l.startswith("s_waitcnt") and "vmcnt" in l and k > loads[0]]
        bad = 0
        for idx, k in enumerate(loads):
            cover = waits[idx // 8]
            assert cover > k, (name, idx)
            dst = regs(body[k].split()[1].rstrip(","))
            for l2 in body[k + 1:cover]:
                touched = set()
                for o in l2.replace(",", " ").split()[1:]:
                    touched |= regs(o)
                if touched & dst:
                    print("HAZARD in %s: %s -> %s" % (name[:48], body[k], l2))
                    bad += 1
        counted = [body[w] for w in waits[:16]]
        ok_counts = counted == ["s_waitcnt vmcnt(24)"] * 14 + ["s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(0)"]
        if not ok_counts:
            print("unexpected wait sequence in %s: %s" % (name[:48], counted))
        report.append((name, bad, ok_counts))
    return report


def main():
    if len(sys.argv) > 1:
        rep = check(sys.argv[1])
    else:
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "vq_assign_filter.s")
            assemble(p)
            rep = check(p)
    for name, bad, okc in rep:
        print("%s: %d hazards, counted waits %s" % (name[:60], bad, "as placed" if okc else "NOT as placed"))
    sys.exit(1 if any(b or not o for _, b, o in rep) else 0)


if __name__ == "__main__":
    main()
