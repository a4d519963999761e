#!/bin/bash
# per-kernel times of the feature-router gate (rocprofv3 --kernel-trace --stats): dual B = 64 / 256, triple B = 128 / 1024
# usage (on the GPU box): bash tools/gate_trace.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/gate_trace}
mkdir -p $R/$OUT
for cfg in "2 64" "2 256" "3 128" "3 1024"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $R/$OUT/n$1_b$2 -o t --output-format csv -- python3 $R/tools/gate_prof.py $1 $2 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/$OUT/n$1_b$2/**/t_kernel_stats.csv", recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if "gate" in r["Name"]:
        print("nb=$1 B=$2", r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1000,2)); tot+=float(r["AverageNs"])/1000
print("nb=$1 B=$2 sum", round(tot,2))
PY
done
