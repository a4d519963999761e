#!/bin/bash
# kernel times of the configs[3] step (triple, feature router) at the per-rank size of N = 8 (B = 128) and at N = 1 (B = 1024)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/strongprof
for b in 128 1024; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/strongprof/b$b -o t --output-format csv -- python3 $R/bench.py --scaling strong --batch $b --steps 100 --warmup 10 --spinup 50 --no-cpu-baseline --no-parity --streams 1 > $R/gpurun_out/strongprof/bench_b$b.json 2>/dev/null
  python3 - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/strongprof/b$b/**/t_kernel_stats.csv", recursive=True)[0]
d=json.loads(open("$R/gpurun_out/strongprof/bench_b$b.json").read().strip().splitlines()[-1])
print("B=$b ms_per_step", round(d["ms_per_step"],4))
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.3: print("  B=$b", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1000,1), r["Percentage"])
PY
done
