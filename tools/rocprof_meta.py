"""After `rocprofv3 --kernel-trace --stats ... -- python3 bench.py <args>`: copy the kernel-stats CSV to
profiles/<name>.csv and write profiles/bench_kernel_stats.meta.json = which sources produced it (bench.source_sha16)
and the dominant kernel's average duration, so that bench.py can quote `roofline.kernel_ms_rocprof` only when the
summary was made from the sources it runs.
  python tools/rocprof_meta.py <rocprof output dir> <dest csv> <path> <scaling> <kernel name prefix> [meta file name]
(default meta file: bench_kernel_stats.meta.json; other bench paths: bench_kernel_stats.<path>.meta.json)"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
d, dest, path, scaling, prefix = sys.argv[1:6]
f = (glob.glob(d + "/*kernel_stats.csv") + glob.glob(d + "/*/*kernel_stats.csv"))[0]
shutil.copy(f, dest)
row = None
for r in csv.DictReader(open(f)):
    if r["Name"].replace("void ", "").startswith(prefix):
        row = r
        break
meta = {"source_sha16": bench.source_sha16(), "path": path, "scaling": scaling, "csv": os.path.relpath(dest, ROOT),
        "dominant_kernel": row["Name"] if row else None,
        "dominant_kernel_avg_ms": float(row["AverageNs"]) * 1e-6 if row else None,
        "dominant_kernel_calls": int(row["Calls"]) if row else None}
mname = sys.argv[6] if len(sys.argv) > 6 else "bench_kernel_stats.meta.json"
json.dump(meta, open(os.path.join(os.path.dirname(dest), mname), "w"), indent=1)
print(json.dumps(meta))
