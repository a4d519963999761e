#!/bin/bash
# Same-box A/B of the round-5 fused step (resolver inside pass 1's launch, self-cleaning workspace) against the round-4
# library (tools/ab/libdvq_r04.so, built from commit 25b4fd7's csrc): serial and 3-stream bench lines + rocprofv3 kernel stats.
# Usage (GPU box): bash tools/ab_fused.sh [outdir under gpurun_out]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-ab_fused}
mkdir -p $O
cd $R
OLD=$R/tools/ab/libdvq_r04.so
for rep in 1 2; do
  DVQ_LIBRARY=$OLD timeout 300 python bench.py --steps 300 --repeats 5 --no-cpu-baseline --no-model-order --no-parity > $O/old_$rep.json 2>> $O/err.log
  timeout 300 python bench.py --steps 300 --repeats 5 --no-cpu-baseline --no-model-order --no-parity > $O/new_$rep.json 2>> $O/err.log
done
cd /tmp && export TMPDIR=/tmp
DVQ_LIBRARY=$OLD timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_old -o t -- python3 $R/bench.py --streams 1 --steps 300 --repeats 3 --no-cpu-baseline --no-parity --no-model-order > $O/trace_old.json 2>> $O/err.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_new -o t -- python3 $R/bench.py --streams 1 --steps 300 --repeats 3 --no-cpu-baseline --no-parity --no-model-order > $O/trace_new.json 2>> $O/err.log
cd $R
python3 - <<PY
import json, glob, csv
O="$O"
for f in sorted(glob.glob(O+"/old_*.json")+glob.glob(O+"/new_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms/step %.4f serial %.4f kernel_ms %.4f whole_op %.4f" % (d["ms_per_step"], d["serial_ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"].get("whole_op_ms", 0)))
    except Exception as e:
        print(f, "ERR", e)
for t in ("old","new"):
    fs=glob.glob(O+"/trace_%s/*kernel_stats.csv"%t)+glob.glob(O+"/trace_%s/*/*kernel_stats.csv"%t)
    if not fs: print(t, "no stats"); continue
    print("==", t)
    for r in list(csv.DictReader(open(fs[0])))[:8]:
        print("  %-90s calls %6s avg %9.1f us  min %9.1f max %9.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
