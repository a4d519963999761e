#!/bin/bash
# timing-only probe of the gate kernel: libdvq variants with a phase removed (NO_BUILD: feature tile not built,
# NO_MFMA: return after the build); per-kernel durations by rocprofv3, configs[3] step at B per rank = $1 (default 128)
B=${1:-128}
cd $GRAFT_REPO_ROOT/dynamicvectorquantization_amd/csrc
cp libdvq.so /tmp/keep.so
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/gateprobe
for v in BASE NO_BUILD NO_MFMA $2; do
  cp $GRAFT_REPO_ROOT/tools/libdvq_$v.so libdvq.so
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gateprobe/rp_$v -o t -- python3 $GRAFT_REPO_ROOT/bench.py --scaling strong --batch $B --streams 1 --no-cpu-baseline --no-parity --steps 100 --spinup 30 > $GRAFT_REPO_ROOT/gpurun_out/gateprobe/bench_$v.log 2>&1)
done
cp /tmp/keep.so libdvq.so
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
for v in "BASE NO_BUILD NO_MFMA $2".split():
    f = glob.glob("gpurun_out/gateprobe/rp_%s/**/*kernel_stats.csv" % v, recursive=True)
    if not f:
        print(v, "no stats", open("gpurun_out/gateprobe/bench_%s.log" % v).read()[-400:]); continue
    for r in csv.DictReader(open(f[0])):
        if "gate" in r["Name"]:
            print("B=$B", v, r["Name"][:30], r["Calls"], round(float(r["AverageNs"]) / 1000, 1))
PY
