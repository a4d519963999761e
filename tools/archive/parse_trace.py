import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(path)))
groups = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    groups.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for name, ds in groups.items():
    if len(ds) >= n and len(ds) % n == 0:
        print(name[:60], " | ".join("%.1f" % (sum(ds[i:i + n][1:]) / (n - 1)) for i in range(0, len(ds), n)), "us")
    else:
        print(name[:60], len(ds), "calls avg %.1f us" % (sum(ds) / len(ds)))
