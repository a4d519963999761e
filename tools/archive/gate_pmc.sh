#!/bin/bash
# PMC passes for the feature-router gate's kernels (separate --pmc runs with --kernel-trace only): HBM bytes (FETCH_SIZE, WRITE_SIZE)
# and SQ counters of gate_pool_kernel / gate_gemm_kernel at triple B = 128 and B = 1024.  usage (GPU box): bash tools/gate_pmc.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 128 1024; do
  i=0
  while read -r line; do
    [ -z "$line" ] && continue
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $line --output-format csv -d $O/b${B}_p$i -o p -- python3 $R/tools/gate_prof.py 3 $B > $O/b${B}_p$i.log 2>&1
  done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
LIST
done
cd $R
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
res = {}
for B in (128, 1024):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(O + "/b%d_p*/**/*counter_collection.csv" % B, recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            short = "pool" if "gate_pool_kernel" in k else ("gemm" if "gate_gemm_kernel" in k else ("finalize" if "gate_finalize" in k else None))
            if short:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(O + "/b%d_p1/**/*kernel_trace.csv" % B, recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            short = "pool" if "gate_pool_kernel" in k else ("gemm" if "gate_gemm_kernel" in k else None)
            if short:
                dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, d in acc.items():
        out[k] = {c: sum(v[5:]) / max(1, len(v[5:])) for c, v in d.items()}
        if k in dur:
            out[k]["duration_us_under_pmc_pass"] = sum(dur[k][5:]) / max(1, len(dur[k][5:]))
    res["triple_B%d" % B] = out
json.dump(res, open(O + "/gate_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
