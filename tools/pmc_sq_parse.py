import csv, glob, collections, sys, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = ("pass1_dense" if "vq_assign_filter_kernel<256, 0, false, false>" in k else
                 "pass1_select_staged_SEL2" if "vq_assign_filter_kernel<256, 2, false, false>" in k else
                 "pass1_select_per_lane_SEL1" if "vq_assign_filter_kernel<256, 1, false, false>" in k else
                 "pass1_conv_fused_CONV" if "vq_assign_filter_kernel<256, 1, true, false>" in k else
                 "pass1_fold_SEL2" if "vq_assign_filter_kernel<256, 2, false, true>" in k else
                 "resolve_fold" if "vq_resolve_kernel<256, true>" in k else
                 ("resolve" if "vq_resolve" in k else ("exact" if "vq_assign_exact" in k else None)))
        if short is None:
            continue
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    # mean per dispatch over the kernel's launches 2-4 of the pass (tools/pmc_workload.py launches every op four times; the
    # later launches of the staged-select kernel are the codes-only / K = 32 variants of the fetch split)
    res[k] = {c: sum(v[1:4]) / max(1, len(v[1:4])) for c, v in d.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k in ("pass1_dense", "pass1_select_staged_SEL2", "pass1_conv_fused_CONV", "pass1_fold_SEL2"):
    print(k)
    for c, v in sorted(res.get(k, {}).items()):
        print("  %-40s %.4g" % (c, v))
