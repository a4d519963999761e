#!/bin/bash
# Same-box A/B of tuning switches (DVQ_TUNE, tuning build) on the serial bench step under rocprofv3.
# Usage (GPU box): bash tools/ab_tune.sh <outdir under gpurun_out> "<tuneA>" "<tuneB>" ["<tuneC>" ...] -- [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; shift
TUNES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do TUNES+=("$1"); shift; done
[ $# -gt 0 ] && shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DVQ_LIBRARY=$R/dynamicvectorquantization_amd/csrc/libdvq_tuning.so
for rep in 0 1; do
  i=0
  for t in "${TUNES[@]}"; do
    DVQ_TUNE="$t" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${i}_$rep -o t -- python3 $R/bench.py --streams 1 --steps 300 --repeats 3 --no-cpu-baseline --no-parity --no-model-order "$@" > $O/trace_${i}_$rep.json 2>> $O/err.log
    i=$((i+1))
  done
done
cd $R
python3 - "$O" "${TUNES[@]}" <<'PY'
import json, glob, csv, sys
O=sys.argv[1]; tunes=sys.argv[2:]
for d in sorted(glob.glob(O+"/trace_*_*")):
    if d.endswith(".json"): continue
    fs=glob.glob(d+"/*kernel_stats.csv")+glob.glob(d+"/*/*kernel_stats.csv")
    if not fs: print(d, "no stats"); continue
    try:
        j=json.loads(open(d+".json").read().strip().splitlines()[-1]); ser=j["serial_ms_per_step"]
    except Exception: ser=None
    i=int(d.split("/")[-1].split("_")[1])
    print("== [%s]" % tunes[i], "serial_ms", ser)
    for r in list(csv.DictReader(open(fs[0])))[:3]:
        print("  %-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
