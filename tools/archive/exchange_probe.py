"""Collect the exchange probe of tools/r03_profiles.sh into one JSON: ms per step with and without the exchange (pack kernel +
RCCL all-gather on a 1-rank group + unpack kernel, two / three exchanges in flight) at 1 and 3 stream slots, and -- from the
kernel trace of the 3-slot run -- the exchange's own kernels and whether RCCL's kernel ran while a pass 1 was running."""
import csv, glob, json, os, sys
O = sys.argv[1]
out = {"note": "1-rank RCCL group on one MI355X: measures the per-step cost of pack + ncclAllGather + unpack and their "
               "interference with pass 1; it is NOT a scaling measurement (no second GPU, no xGMI traffic)"}
for S in (1, 3):
    try:
        a = json.load(open(os.path.join(O, "xch_plain_s%d.json" % S)))
        b = json.load(open(os.path.join(O, "xch_rccl1_s%d.json" % S)))
        out["streams_%d" % S] = {"ms_per_step_plain": a["ms_per_step"], "ms_per_step_with_exchange": b["ms_per_step"],
                                 "added_us_per_step": (b["ms_per_step"] - a["ms_per_step"]) * 1e3}
    except Exception as e:
        out["streams_%d" % S] = {"error": repr(e)}
try:
    f = (glob.glob(O + "/trace_xch/*kernel_trace.csv") + glob.glob(O + "/trace_xch/*/*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    ks = {}
    p1 = []
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ks.setdefault(n, []).append((s, e))
        if n.startswith("vq_assign_filter_kernel"):
            p1.append((s, e))
    p1.sort()
    def overlap(iv):
        import bisect
        starts = [s for s, _ in p1]
        tot = 0
        for s, e in iv:
            i = max(0, bisect.bisect_left(starts, s) - 2)
            while i < len(p1) and p1[i][0] < e:
                tot += max(0, min(e, p1[i][1]) - max(s, p1[i][0]))
                i += 1
        return tot
    ex = {}
    for n, iv in ks.items():
        if n.startswith(("xch_", "ncclDevKernel", "ncclKernel")) or "nccl" in n.lower() or "rccl" in n.lower():
            d = [e - s for s, e in iv]
            ex[n[:80]] = {"calls": len(iv), "avg_us": sum(d) / len(d) / 1e3,
                          "fraction_of_its_time_under_a_running_pass1": overlap(iv) / max(1, sum(d))}
    out["exchange_kernels_3_streams"] = ex
    d = [e - s for s, e in p1]
    out["pass1_avg_us_in_that_trace"] = sum(d) / max(1, len(d)) / 1e3
except Exception as e:
    out["trace_error"] = repr(e)
print(json.dumps(out, indent=1))
