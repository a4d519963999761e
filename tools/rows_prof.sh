#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/rowsprof
for form in 0 2; do for what in codes zq; do
  DVQ_ROUTED_DEDUP=$form rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rowsprof/f${form}_$what -o t --output-format csv -- python3 $R/tools/rows_prof.py $what > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/rowsprof/f${form}_$what/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.5: print("form $form $what", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
done; done
