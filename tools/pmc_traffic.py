"""Turn the two PMC passes (FETCH_SIZE, WRITE_SIZE; units: KiB... 1024-byte units per the guide)
into profiles/pmc_traffic.json: HBM bytes per launch of the filter kernel, with the gfx950 FETCH_SIZE
correction calibrated on the exact kernel's known z read (same 4-B-per-lane access pattern)."""
import csv, glob, json, sys, collections
def load(d, counter):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return per
fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
N, D = 256 * 1024, 256
known_read = N * D * 4 + 32 * (D * 4 + D * 4 + 8)            # z + the 32-code f32 tile image
cal_key = [k for k in fetch if k.startswith("vq_assign_exact_kernel<256, false, false")][0]   # dense instantiation
cal = [v for v in fetch[cal_key][:4]]
cal_kib = sum(cal[1:]) / len(cal[1:])
factor = known_read / (cal_kib * 1024.0)
out = {"calibration": {"kernel": "vq_assign_exact_kernel<256, false> K=32 codes-only", "known_read_bytes": known_read,
                       "FETCH_SIZE_raw_bytes": cal_kib * 1024.0, "fetch_correction_factor": factor}}
def avg(per, name):
    v = per[name][:4]            # launches 2-4 of the kernel's FIRST group of four (tools/pmc_workload.py launches every op four times;
    return sum(v[1:]) / max(1, len(v[1:]))   # later groups of the staged-select kernel are the variants of the fetch split below)
kernels = {}
def find(per, prefix):
    ks = [k for k in per if k.startswith(prefix)]
    return ks[0] if ks else None
for label, prefix in (("dense_pass1", "vq_assign_filter_kernel<256, 0, false, false>"), ("fused_pass1", "vq_assign_filter_kernel<256, 2, false, false>"),
                      ("fused_pass1_per_lane_select", "vq_assign_filter_kernel<256, 1, false, false>"),
                      ("model_pass1", "vq_assign_filter_kernel<256, 1, true, false>"), ("fold_pass1", "vq_assign_filter_kernel<256, 2, false, true>"),
                      ("resolver", "vq_resolve_kernel<256, false>"), ("fold_resolver", "vq_resolve_kernel<256, true>")):
    kf, kw = find(fetch, prefix), find(write, prefix)
    if kf is None or kw is None:
        continue
    f_raw, w = avg(fetch, kf) * 1024.0, avg(write, kw) * 1024.0
    kernels[label] = {"kernel": kf, "FETCH_SIZE_raw_bytes": f_raw, "fetch_bytes_corrected": f_raw * factor, "write_bytes": w,
                      "hbm_bytes_per_launch": f_raw * factor + w}
out["kernels"] = kernels
alg = N * (D * 4 * 2 + 8 + 4) + 1024 * D * 4
if "dense_pass1" in kernels:
    out["filter"] = {"hbm_bytes_per_launch": kernels["dense_pass1"]["hbm_bytes_per_launch"], "algorithmic_bytes": alg,
                     "ratio_dominant_kernel_to_algorithmic": kernels["dense_pass1"]["hbm_bytes_per_launch"] / alg}
if "fused_pass1" in kernels:
    # the fused kernel reads every line of h_fine and of h_coarse once (a 128-B line of h_fine holds 16 cells: all of them are
    # needed) and writes z_q once: line-granular floor = N*1024 (fine) + N*256 (coarse) + z_q + codes + mask + gate + codebook
    floor = N * (D * 4 + D + D * 4 + 8 + 4) + N // 4 * 4 + 1024 * D * 4
    out["routed"] = {"hbm_bytes_per_launch": kernels["fused_pass1"]["hbm_bytes_per_launch"], "algorithmic_bytes": alg,
                     "ratio_dominant_kernel_to_algorithmic": kernels["fused_pass1"]["hbm_bytes_per_launch"] / alg,
                     "line_granular_floor_bytes": floor,
                     "ratio_to_line_granular_floor": kernels["fused_pass1"]["hbm_bytes_per_launch"] / floor,
                     "note": "resolver kernels of both ops are averaged together under 'resolver'"}
    if "fused_pass1_per_lane_select" in kernels:
        out["routed"]["per_lane_select_form_bytes"] = kernels["fused_pass1_per_lane_select"]["hbm_bytes_per_launch"]
if "model_pass1" in kernels:
    # select + 1x1 quant_conv + assign as one kernel: same algorithmic bytes as the routed op (the conv's output is never written);
    # the 256 KiB of weight images are re-read by every workgroup from L2
    floor = N * (D * 4 + D + D * 4 + 8 + 4) + N // 4 * 4 + 1024 * D * 4
    out["model"] = {"hbm_bytes_per_launch": kernels["model_pass1"]["hbm_bytes_per_launch"], "algorithmic_bytes": alg,
                    "ratio_dominant_kernel_to_algorithmic": kernels["model_pass1"]["hbm_bytes_per_launch"] / alg,
                    "line_granular_floor_bytes": floor,
                    "ratio_to_line_granular_floor": kernels["model_pass1"]["hbm_bytes_per_launch"] / floor,
                    "note": "per-lane select form (the ring slots the staged form parks the coarse branch in carry the conv's weights)"}
# the routed pass 1's fetch split by what it is.  Dispatches of vq_assign_filter_kernel<256, 2, false, false>
# in tools/pmc_workload.py order: [0:4] the whole op (3), then -- after the per-lane-select runs, a different instantiation --
# [4:8] pass 1 alone (6), [8:12] codes only (7: no e-row gather), [12:16] codes only against K = 32 (8: one code tile)
kf = find(fetch, "vq_assign_filter_kernel<256, 2, false, false>")
kw = find(write, "vq_assign_filter_kernel<256, 2, false, false>")
if kf is not None and len(fetch[kf]) >= 16:
    grp = lambda per, k, i: sum(per[k][4 * i + 1:4 * i + 4]) / 3.0 * 1024.0
    full, codes_only, k32 = (grp(fetch, kf, i) * factor for i in (1, 2, 3))
    branch_lines = N * (D * 4 + D)                      # every 128-B line of h_fine and of h_coarse once
    out["routed_fetch_split"] = {
        "pass1_full_fetch_bytes": full, "pass1_codes_only_fetch_bytes": codes_only, "pass1_codes_only_K32_fetch_bytes": k32,
        "e_row_gather_fetch_bytes": full - codes_only, "code_image_refetch_bytes": codes_only - k32,
        "branch_line_floor_bytes": branch_lines, "K32_fetch_over_branch_lines": k32 / branch_lines,
        "write_bytes": {"full": grp(write, kw, 1), "codes_only": grp(write, kw, 2), "codes_only_K32": grp(write, kw, 3)} if kw else None,
        "note": "fetch counters corrected by the calibration factor; e-row gather = full - codes only; code-image / seed "
                "re-fetch from the Infinity Cache = codes only - (codes only at K = 32); what remains above the branch-line "
                "floor at K = 32 is gate / uncalibrated 16-B DMA counting"}
if "fold_pass1" in kernels:
    floor = N * (D * 4 + D + D * 4 + 8 + 4) + N // 4 * 4 + 1024 * D * 4
    out["model_fold"] = {"hbm_bytes_per_launch": kernels["fold_pass1"]["hbm_bytes_per_launch"], "algorithmic_bytes": alg,
                         "ratio_dominant_kernel_to_algorithmic": kernels["fold_pass1"]["hbm_bytes_per_launch"] / alg,
                         "line_granular_floor_bytes": floor}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("filter", "routed", "model", "model_fold", "routed_fetch_split")}), "factor", factor)
