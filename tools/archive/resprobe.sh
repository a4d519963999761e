#!/bin/bash
# timing-only probe of the resolver: libdvq variants with a phase removed (tools/libdvq_x1.so: no exact chains,
# x2: no candidate enumeration); per-kernel durations by rocprofv3
cd $GRAFT_REPO_ROOT/dynamicvectorquantization_amd/csrc
cp libdvq.so /tmp/keep.so
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/resprobe
for v in A P W A P W; do
  cp $GRAFT_REPO_ROOT/tools/libdvq_$v.so libdvq.so
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/resprobe/rp_$v -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --steps 150 --spinup 20 > $GRAFT_REPO_ROOT/gpurun_out/resprobe/bench_$v.log 2>&1)
done
cp /tmp/keep.so libdvq.so
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for v in ("A", "P", "W"):
    f = glob.glob("gpurun_out/resprobe/rp_%s/*kernel_stats.csv" % v) + glob.glob("gpurun_out/resprobe/rp_%s/*/*kernel_stats.csv" % v)
    if not f:
        print(v, "no stats", open("gpurun_out/resprobe/bench_%s.log" % v).read()[-400:]); continue
    for r in csv.DictReader(open(f[0])):
        if "resolve" in r["Name"] or "filter_kernel" in r["Name"]:
            print(v, r["Name"][:36], r["Calls"], round(float(r["AverageNs"]) / 1000, 1))
PY
