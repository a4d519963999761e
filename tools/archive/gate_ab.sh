#!/bin/bash
# rocprofv3 kernel stats of the feature-router gate op for several builds of the library (tuning variants), same box.
# usage: bash tools/gate_ab.sh <outdir> "<nb B> ..." lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; CFGS="$2"; shift 2
mkdir -p $O
for rep in 1 2; do for L in "$@"; do
  export DVQ_LIBRARY=$R/dynamicvectorquantization_amd/csrc/$L
  IFS=';' read -ra CF <<< "$CFGS"
  for cfg in "${CF[@]}"; do
    set -- $cfg
    d=$O/${L%.so}_n$1_b$2_$rep
    rocprofv3 --kernel-trace --stats -d $d -o t --output-format csv -- python3 $R/tools/gate_prof.py $1 $2 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("$d/**/t_kernel_stats.csv", recursive=True)
if f:
    tot=0; out=[]
    for r in csv.DictReader(open(f[0])):
        if "gate" in r["Name"]:
            out.append("%s %.1f" % (r["Name"].split("(")[0].replace("void ","")[:30], float(r["AverageNs"])/1000)); tot+=float(r["AverageNs"])/1000
    print("$L nb=$1 B=$2 rep=$rep |", " | ".join(out), "| sum %.1f" % tot)
PY
  done
  set -- "$@"
done; done
