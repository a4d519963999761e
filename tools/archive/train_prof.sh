#!/bin/bash
# rocprofv3 kernel stats of the training-mode step (tools/train_step_probe.py); GPU box.  usage: bash tools/train_prof.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-train_prof}; mkdir -p $O
python3 $R/tools/train_step_probe.py 2>/dev/null | tee $O/train_step.jsonl
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/train_step_probe.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace/**/t_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-95s calls %5s avg %8.1f us  total %5.1f%%" % (r["Name"][:95], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
