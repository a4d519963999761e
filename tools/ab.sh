#!/bin/bash
# A/B two builds of libdvq.so on the same box: usage tools/ab.sh A.so B.so [runs] [extra bench args]
A=$1; B=$2; R=${3:-3}; shift 3
cd $GRAFT_REPO_ROOT/dynamicvectorquantization_amd/csrc
cp libdvq.so /tmp/libdvq_keep.so
for r in $(seq 1 $R); do
  for v in A B; do
    f=$A; [ $v = B ] && f=$B
    cp $f libdvq.so
    (cd ../..; python bench.py --no-cpu-baseline --no-parity --steps 200 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_ms']*1000,1), round(d['roofline']['whole_op_ms']*1000,1))")
  done
done
cp /tmp/libdvq_keep.so libdvq.so
