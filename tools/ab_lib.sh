#!/bin/bash
# Same-box A/B of two builds of the library under rocprofv3 (kernel stats of bench.py --streams 1) + the serial bench line.
# Usage (GPU box): bash tools/ab_lib.sh <outdir under gpurun_out> <libA.so> <libB.so> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; A=$2; B=$3; shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in A B A B; do
  L=$A; [ $t = B ] && L=$B
  n=$(ls $O | grep -c "^trace_$t") 
  DVQ_LIBRARY=$R/$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${t}_$n -o t -- python3 $R/bench.py --streams 1 --steps 300 --repeats 3 --no-cpu-baseline --no-parity --no-model-order "$@" > $O/trace_${t}_$n.json 2>> $O/err.log
done
cd $R
python3 - <<PY
import json, glob, csv
O="$O"
for d in sorted(glob.glob(O+"/trace_*_*")):
    if d.endswith(".json"): continue
    fs=glob.glob(d+"/*kernel_stats.csv")+glob.glob(d+"/*/*kernel_stats.csv")
    if not fs: print(d, "no stats"); continue
    try:
        j=json.loads(open(d+".json").read().strip().splitlines()[-1]); ser=j["serial_ms_per_step"]
    except Exception: ser=None
    print("==", d.split("/")[-1], "serial_ms", ser)
    for r in list(csv.DictReader(open(fs[0])))[:5]:
        print("  %-70s calls %6s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
