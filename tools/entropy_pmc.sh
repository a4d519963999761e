#!/bin/bash
# SQ counter passes for entropy_map_kernel at B = 256 (separate --pmc runs, kernel-trace only; SQ counters only).
# usage (on the GPU box): bash tools/entropy_pmc.sh <outdir>
OUT=${1:-gpurun_out/ent_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $line --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/entropy_pmc_workload.py > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
FETCH_SIZE
GRBM_GUI_ACTIVE SQ_WAVES
LIST
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
acc = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "entropy_map_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in sorted(glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "entropy_map_kernel" in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
res = {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in acc.items()}
res["kernel_us_under_pmc"] = sum(dur[2:]) / max(1, len(dur[2:]))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
