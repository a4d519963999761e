import csv, glob, collections, sys, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = ("pass1_fused_select" if "vq_assign_filter_kernel<256, 1, false>" in k else
                 "filter" if "vq_assign_filter" in k else
                 ("resolve" if "vq_resolve" in k else ("exact" if "vq_assign_exact" in k else None)))
        if short is None:
            continue
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}     # mean per dispatch
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k in ("filter", "pass1_fused_select"):
    print(k)
    for c, v in sorted(res.get(k, {}).items()):
        print("  %-40s %.4g" % (c, v))
