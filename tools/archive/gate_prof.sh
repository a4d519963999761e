#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DVQ_LIBRARY=$R/dynamicvectorquantization_amd/csrc/libdvq_tuning.so   # the scalar / vector pooling switch exists in the tuning build only
mkdir -p $R/gpurun_out/gateprof
for sc in 0 1 0 1; do for cfg in "2 64" "2 256" "3 128" "3 512"; do
  set -- $cfg
  export DVQ_GATE_POOL_SCALAR=$sc
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gateprof/n$1_b$2 -o t --output-format csv -- python3 $R/tools/gate_prof.py $1 $2 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/gateprof/n$1_b$2/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gate" in r["Name"] or "w1_split" in r["Name"]: print("scalar=$sc nb=$1 B=$2", r["Name"][:34], r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
done; done
